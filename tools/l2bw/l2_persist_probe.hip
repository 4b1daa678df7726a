// Does what a kernel pulled into its XCD's L2 survive the kernel boundary?  (The premise of prefetching a share of the NEXT decode-attention launch's K/V rows from the
// fused MLP launch, whose 11 us leave the fabric idle: rows parked in the memory-side cache are no faster than HBM - tools/mall_probe.py - but rows in the XCD's own L2 would
// not cross the fabric at all.)
//   hipcc -O3 --offload-arch=gfx950 -o l2_persist_probe l2_persist_probe.hip && ./l2_persist_probe [region_KiB=96] [nt_reader=1]
// 256 workgroups x 1024 threads (workgroup i runs on XCD i % 8).  `touch` makes workgroup i read region (i + shift) % 256 with plain loads or LDS-DMA; `timed` makes
// workgroup i read region i (non-temporal loads, like the decode walk) and records its own duration (s_memrealtime, 100 MHz).  Cases:
//   same XCD, previous kernel      touch(shift 0) ; timed         -> an L2 hit if L2 contents survive the boundary
//   other XCD, previous kernel     touch(shift 1) ; timed         -> memory-side cache hit at best
//   cold                           flush (1 GiB streamed) ; timed -> HBM
//   same kernel                    timed reads the region twice, second pass reported (the L2-hit reference)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>   // 0 plain loads, 1 LDS-DMA
__global__ __launch_bounds__(1024) void touch_kernel(const f32x4* __restrict__ buf, float* sink, long region_f4, int shift) {
    __shared__ __attribute__((aligned(1024))) f32x4 lds[16 * 64];
    const long region = (blockIdx.x + shift) % gridDim.x;
    const f32x4* src = buf + region * region_f4;
    const int tid = threadIdx.x, wave = tid >> 6;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (long i = tid; i < region_f4; i += 1024) {
        if (MODE == 0) acc += src[i];
        else __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + i), (__attribute__((address_space(3))) void*)(lds + wave * 64), 16, 0, 0);
    }
    if (MODE == 1) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); acc += lds[tid]; }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) sink[blockIdx.x] = acc[0];
}

template <bool NT>
__global__ __launch_bounds__(1024) void timed_kernel(const f32x4* __restrict__ buf, float* sink, long region_f4, long long* t_out, int passes) {
    const f32x4* src = buf + (long)blockIdx.x * region_f4;
    const int tid = threadIdx.x;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int p = 0; p < passes; ++p) {
        __syncthreads();
        const long long t0 = __builtin_amdgcn_s_memrealtime();
        for (long i = tid; i < region_f4; i += 4096) {
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long j = i + u * 1024 < region_f4 ? i + u * 1024 : i;
                v[u] = NT ? __builtin_nontemporal_load(src + j) : src[j];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) acc += v[u];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const long long t1 = __builtin_amdgcn_s_memrealtime();
        if (tid == 0) t_out[(long)p * gridDim.x + blockIdx.x] = t1 - t0;
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) sink[blockIdx.x] = acc[0];
}

// coherence across the boundary: workgroup i OVERWRITES region (i + shift) % n with `val`; check_kernel counts the words of region i that differ from `val`
__global__ __launch_bounds__(1024) void write_kernel(f32x4* __restrict__ buf, long region_f4, int shift, float val) {
    f32x4* dst = buf + (long)((blockIdx.x + shift) % gridDim.x) * region_f4;
    for (long i = threadIdx.x; i < region_f4; i += 1024) dst[i] = f32x4{val, val, val, val};
}
__global__ __launch_bounds__(1024) void check_kernel(const f32x4* __restrict__ buf, long region_f4, float val, unsigned* bad) {
    const f32x4* src = buf + (long)blockIdx.x * region_f4;
    unsigned n = 0;
    for (long i = threadIdx.x; i < region_f4; i += 1024) {
        const f32x4 v = __builtin_nontemporal_load(src + i);
        n += (v[0] != val) + (v[1] != val) + (v[2] != val) + (v[3] != val);
    }
    if (n) atomicAdd(bad, n);
}

__global__ __launch_bounds__(256) void flush_kernel(const f32x4* __restrict__ buf, float* sink, long n) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) acc += buf[i];
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) sink[blockIdx.x] = acc[0];
}

int main(int argc, char** argv) {
    const int region_kb = argc > 1 ? atoi(argv[1]) : 96;
    const bool nt = argc > 2 ? atoi(argv[2]) != 0 : true;
    const int blocks = 256;
    const long region_f4 = (long)region_kb * 1024 / 16;
    f32x4 *buf, *big; float* sink; long long* t_d;
    const long big_n = (1L << 30) / 16;
    CK(hipMalloc(&buf, blocks * region_f4 * 16)); CK(hipMemset(buf, 0, blocks * region_f4 * 16));
    CK(hipMalloc(&big, big_n * 16)); CK(hipMemset(big, 0, big_n * 16));
    CK(hipMalloc(&sink, 1 << 20)); CK(hipMalloc(&t_d, 2 * blocks * sizeof(long long)));
    std::vector<long long> t_h(2 * blocks);
    auto timed = [&](int passes) {
        if (nt) hipLaunchKernelGGL(timed_kernel<true>, dim3(blocks), dim3(1024), 0, 0, buf, sink, region_f4, t_d, passes);
        else hipLaunchKernelGGL(timed_kernel<false>, dim3(blocks), dim3(1024), 0, 0, buf, sink, region_f4, t_d, passes);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(t_h.data(), t_d, 2 * blocks * sizeof(long long), hipMemcpyDeviceToHost));
    };
    auto report = [&](const char* what, int pass) {
        std::vector<double> us(blocks);
        for (int i = 0; i < blocks; ++i) us[i] = t_h[pass * blocks + i] / 100.0;
        std::sort(us.begin(), us.end());
        double mean = 0; for (double x : us) mean += x; mean /= blocks;
        printf("%-44s per workgroup: mean %6.2f us  median %6.2f  max %6.2f   -> %6.1f GB/s per CU, %5.2f TB/s over the chip\n", what, mean, us[blocks / 2], us[blocks - 1],
               region_kb * 1024.0 / (mean * 1e-6) / 1e9, blocks * region_kb * 1024.0 / (mean * 1e-6) / 1e12);
    };
    auto flush = [&]() { hipLaunchKernelGGL(flush_kernel, dim3(2048), dim3(256), 0, 0, big, sink, big_n); };
    printf("region %d KiB per workgroup (%.1f MiB per XCD), reader loads %s\n", region_kb, 32.0 * region_kb / 1024.0, nt ? "non-temporal" : "plain");
    for (int rep = 0; rep < 2; ++rep) {
        for (int mode = 0; mode < 2; ++mode) {
            flush();
            if (mode == 0) hipLaunchKernelGGL(touch_kernel<0>, dim3(blocks), dim3(1024), 0, 0, buf, sink, region_f4, 0);
            else hipLaunchKernelGGL(touch_kernel<1>, dim3(blocks), dim3(1024), 0, 0, buf, sink, region_f4, 0);
            timed(1); report(mode == 0 ? "same XCD, previous kernel (plain loads)" : "same XCD, previous kernel (LDS-DMA)", 0);
            flush();
            if (mode == 0) hipLaunchKernelGGL(touch_kernel<0>, dim3(blocks), dim3(1024), 0, 0, buf, sink, region_f4, 1);
            else hipLaunchKernelGGL(touch_kernel<1>, dim3(blocks), dim3(1024), 0, 0, buf, sink, region_f4, 1);
            timed(1); report(mode == 0 ? "other XCD, previous kernel (plain loads)" : "other XCD, previous kernel (LDS-DMA)", 0);
        }
        flush(); timed(2);
        report("cold (1 GiB streamed in between)", 0);
        report("same kernel, second pass", 1);
    }
    // a region cached (clean) in XCD x's L2 by one kernel, overwritten from ANOTHER XCD by the next, read again on XCD x by a third: stale words seen?
    unsigned* bad; CK(hipMalloc(&bad, 4)); CK(hipMemset(bad, 0, 4));
    for (int rep = 0; rep < 4; ++rep) {
        const float val = 1.0f + rep;
        hipLaunchKernelGGL(touch_kernel<0>, dim3(blocks), dim3(1024), 0, 0, buf, sink, region_f4, 0);
        hipLaunchKernelGGL(write_kernel, dim3(blocks), dim3(1024), 0, 0, buf, region_f4, 1 + rep, val);
        hipLaunchKernelGGL(check_kernel, dim3(blocks), dim3(1024), 0, 0, buf, region_f4, val, bad);
    }
    unsigned bad_h = 0; CK(hipMemcpy(&bad_h, bad, 4, hipMemcpyDeviceToHost));
    printf("coherence across kernel boundaries (cached on XCD x, overwritten from another XCD, read on XCD x): %u stale words\n", bad_h);
    return 0;
}
