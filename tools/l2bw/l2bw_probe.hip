// L2 -> CU read bandwidth of the whole chip (all 256 CUs reading L2-resident data), the ceiling both Route M throughput kernels are priced against in DESIGN.md.
//   hipcc -O3 --offload-arch=gfx950 -o l2bw_probe l2bw_probe.hip && ./l2bw_probe
// Every workgroup streams a REGION of `region_kb` KiB again and again (larger than the CU's 32 KiB vector L1, so every pass comes from L2); workgroups are dispatched
// round-robin over the 8 XCDs (block b -> XCD b % 8), and the `share` workgroups b, b + 8, b + 16, ... of one XCD read the SAME region (like a weight tile or a bias
// image shared by the workgroups of an XCD), so the footprint per XCD is (blocks per XCD / share) regions.  Variants: plain 16-byte loads to registers, and LDS-DMA
// (global_load_lds_dwordx4), the path the GEMM uses.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int MODE, int U>   // MODE 0: global_load_dwordx4 -> VGPR, 1: global_load_lds_dwordx4 -> LDS; U loads of 16 B per thread in flight per iteration
__global__ __launch_bounds__(512) void l2_read_kernel(const float4* __restrict__ buf, float* __restrict__ sink, long region_f4, int share, int passes) {
    __shared__ __attribute__((aligned(1024))) float4 lds[8 * 64 * U];   // 8 waves x U KiB landing zone
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const long region = (long)xcd * 64 + slot / share;                 // region id: distinct per XCD
    const float4* src = buf + region * region_f4;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const long n_iter = region_f4 / (512 * U);
    const long start = ((long)(slot % share) * 977) % n_iter;            // the sharers walk the region out of phase
    for (int p = 0; p < passes; ++p) {
        for (long it = 0; it < n_iter; ++it) {
            const long i = ((it + start) % n_iter) * (512 * U) + tid;
            if (MODE == 0) {
                float4 v[U];
#pragma unroll
                for (int u = 0; u < U; ++u) v[u] = src[i + u * 512];
#pragma unroll
                for (int u = 0; u < U; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
            } else {
#pragma unroll
                for (int u = 0; u < U; ++u)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + i + u * 512),
                                                     (__attribute__((address_space(3))) void*)(lds + wave * (64 * U) + u * 64), 16, 0, 0);
            }
        }
        if (MODE == 1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            acc.x += lds[tid & (512 * U - 1)].x;
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) sink[blockIdx.x] = acc.x;
}

int main(int argc, char** argv) {
    const int passes = argc > 1 ? atoi(argv[1]) : 40;
    float* sink; CK(hipMalloc(&sink, 1 << 20));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("mode region_KiB share blocks  footprint/XCD_MiB   TB/s   B/clk/CU@2.4GHz\n");
    // mode 0 / 2 / 3: plain loads, 4 / 8 / 16 in flight per thread; 1 / 4: LDS-DMA, 4 / 8 per wave-instruction batch
    for (int mode = 0; mode < 5; ++mode)
        for (int region_kb : {128, 2048})
            for (int share : {4, 32})
                for (int blocks : {256, 512}) {
                    const long region_f4 = (long)region_kb * 1024 / 16;
                    const int per_xcd = blocks / 8, regions_per_xcd = (per_xcd + share - 1) / share;
                    const double foot = (double)regions_per_xcd * region_kb / 1024.0;
                    if (foot > 3.0) continue;                          // must stay inside the XCD's 4 MiB L2
                    float4* buf; const size_t n = ((size_t)7 * 64 + regions_per_xcd) * region_f4;
                    CK(hipMalloc(&buf, n * sizeof(float4))); CK(hipMemset(buf, 0, n * sizeof(float4)));
                    auto launch = [&](int p) {
                        if (mode == 0) hipLaunchKernelGGL((l2_read_kernel<0, 4>), dim3(blocks), dim3(512), 0, 0, buf, sink, region_f4, share, p);
                        else if (mode == 1) hipLaunchKernelGGL((l2_read_kernel<1, 4>), dim3(blocks), dim3(512), 0, 0, buf, sink, region_f4, share, p);
                        else if (mode == 2) hipLaunchKernelGGL((l2_read_kernel<0, 8>), dim3(blocks), dim3(512), 0, 0, buf, sink, region_f4, share, p);
                        else if (mode == 3) hipLaunchKernelGGL((l2_read_kernel<0, 16>), dim3(blocks), dim3(512), 0, 0, buf, sink, region_f4, share, p);
                        else hipLaunchKernelGGL((l2_read_kernel<1, 8>), dim3(blocks), dim3(512), 0, 0, buf, sink, region_f4, share, p);
                    };
                    const int p_eff = passes * (2048 / region_kb);    // same bytes per block for every region size
                    launch(2); CK(hipDeviceSynchronize());
                    CK(hipEventRecord(e0)); launch(p_eff); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    const double total = (double)blocks * p_eff * region_kb * 1024.0;
                    printf("%4d %10d %5d %6d %18.2f %7.2f %10.1f\n", mode, region_kb, share, blocks, foot, total / (ms * 1e-3) / 1e12, total / (ms * 1e-3) / 256 / 2.4e9);
                    fflush(stdout);
                    CK(hipFree(buf));
                }
    return 0;
}
