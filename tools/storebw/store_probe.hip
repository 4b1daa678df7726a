// Store-pattern probe: 256 x 128 fp32 tiles of a [M, 1024] matrix written by 8-wave workgroups (64 x 64 patch per wave, sixteen 16-byte stores per lane), the way the
// LDS-DMA GEMM's epilogue does it.  Pattern A = the MFMA accumulator layout (lane -> row lane & 31, 32 bytes per row and instruction: 32 partial lines per instruction);
// pattern B = row-major after an LDS transpose (16 lanes = 256 contiguous bytes of one row, 4 rows per instruction: 8 whole lines per instruction).
// build: hipcc -O3 --offload-arch=gfx950 store_probe.hip -o store_probe ; run: ./store_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int PAT>
__global__ __launch_bounds__(512) void k(float* out, int ld, int tiles_x, int ntiles, unsigned long long* cyc) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
    unsigned long long t_acc = 0;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int ty = t / tiles_x, tx = t - ty * tiles_x;
        float* base = out + (long)(ty * 256 + wm * 64) * ld + tx * 128 + wn * 64;
        const f32x4 v = {(float)t, (float)lane, 1.f, 2.f};
        const unsigned long long t0 = __builtin_readcyclecounter();
        if (PAT == 0) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq)
                        *reinterpret_cast<f32x4*>(base + (long)(i * 32 + (lane & 31)) * ld + j * 32 + 8 * qq + 4 * (lane >> 5)) = v;
        } else if (PAT == 1) {
#pragma unroll
            for (int s = 0; s < 16; ++s) *reinterpret_cast<f32x4*>(base + (long)(s * 4 + (lane >> 4)) * ld + (lane & 15) * 4) = v;
        } else {   // 8 lanes = 128 contiguous bytes, 8 rows per instruction
#pragma unroll
            for (int s = 0; s < 16; ++s) *reinterpret_cast<f32x4*>(base + (long)((s >> 1) * 8 + (lane >> 3)) * ld + (s & 1) * 32 + (lane & 7) * 4) = v;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        t_acc += __builtin_readcyclecounter() - t0;
    }
    if (threadIdx.x == 0) cyc[blockIdx.x] = t_acc;
}
int main() {
    const int M = 24576, N = 1024, tiles_x = N / 128, ntiles = (M / 256) * tiles_x;
    float* out; unsigned long long* cyc;
    CK(hipMalloc(&out, (size_t)M * N * 4)); CK(hipMalloc(&cyc, 256 * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int G : {256, 64, 16})
    for (int pat = 0; pat < 3; ++pat) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            if (pat == 0) hipLaunchKernelGGL(k<0>, dim3(G), dim3(512), 0, 0, out, N, tiles_x, ntiles, cyc);
            else if (pat == 1) hipLaunchKernelGGL(k<1>, dim3(G), dim3(512), 0, 0, out, N, tiles_x, ntiles, cyc);
            else hipLaunchKernelGGL(k<2>, dim3(G), dim3(512), 0, 0, out, N, tiles_x, ntiles, cyc);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            unsigned long long h[256]; CK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
            double avg = 0; for (int i = 0; i < G; ++i) avg += h[i]; avg /= (double)G * (ntiles / G);
            printf("%3d workgroups, pattern %d: %.1f us for %d tiles (%.2f TB/s), %.0f cycles per tile store phase (128 KB per CU: %.1f B/cyc/CU)\n", G, pat, ms * 1e3, ntiles,
                   (double)M * N * 4 / (ms * 1e-3) / 1e12, avg, 131072.0 / avg);
            printf("      = %.1f GB/s per active CU\n", (double)M * N * 4 / (ms * 1e-3) / 1e9 / G);
        }
    }
    return 0;
}
