// Micro-benchmark for the persistent Route-A decode-step kernel: what does a grid-wide barrier cost on MI355X when 256 workgroups of 1024 threads
// (one per CU) exchange a few KB through global memory with per-access agent-scope loads / stores (no bulk L2 write-back / invalidate)?
//   ./gridbar_probe [rounds] [mode]   mode 0: scoped accesses + counter barrier, 1: plain accesses + __threadfence() around the barrier
// Every spin is bounded: a barrier that does not complete sets an error flag and the kernel drains (never hangs the box).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ void st_agent(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_agent(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

struct GridBar {
    unsigned* counter;   // monotonically increasing arrivals
    unsigned* error;
    unsigned nwg;
    unsigned gen;        // barriers passed so far (per thread copy)
};

__device__ __forceinline__ void grid_barrier(GridBar& b, bool fence) {
    __builtin_amdgcn_s_waitcnt(0);   // every earlier store of this wave has been acknowledged
    if (fence) __threadfence();
    __syncthreads();
    b.gen += 1;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(b.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = b.gen * b.nwg;
        unsigned spins = 0;
        while (__hip_atomic_load(b.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 22) || __hip_atomic_load(b.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                __hip_atomic_store(b.error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
    }
    __syncthreads();
    if (fence) __threadfence();
}

__global__ __launch_bounds__(1024) void probe_kernel(float* xch, unsigned* counter, unsigned* error, unsigned* mism, int rounds, int mode, const float* wsrc, float* sink) {
    extern __shared__ float lds[];
    GridBar b{counter, error, gridDim.x, 0};
    const int wg = blockIdx.x, tid = threadIdx.x, nwg = gridDim.x;
    unsigned bad = 0;
    float keep = 0.f;
    for (int r = 0; r < rounds; ++r) {
        float* slot = xch + ((size_t)(r & 1) * nwg + wg) * 64;
        // a 64 KB weight slice lands in LDS by DMA while the round runs (as the real kernel prefetches its next weights)
        if (wsrc) {
            const int wave = tid >> 6, lane = tid & 63;
            for (int j = 0; j < 4; ++j) {
                const float* src = wsrc + ((size_t)wg * 16384 + (size_t)(wave * 4 + j) * 256 + lane * 4);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)(lds + (wave * 4 + j) * 256), 16, 0, 0);
            }
        }
        if (tid < 64) {
            const float v = (float)(r * 1000 + wg) + 0.001f * tid;
            if (mode == 0) st_agent(slot + tid, v); else slot[tid] = v;
        }
        grid_barrier(b, mode == 1);
        // read 16 other workgroups' slots
        const int j = tid >> 6, l = tid & 63;
        const int src = (wg + 1 + j * 17) % nwg;
        const float* p = xch + ((size_t)(r & 1) * nwg + src) * 64 + l;
        const float got = mode == 0 ? ld_agent(p) : *p;
        const float want = (float)(r * 1000 + src) + 0.001f * l;
        if (got != want) bad += 1;
        if (wsrc) keep += lds[(tid * 4) & 16383];
    }
    if (bad) atomicAdd(mism, bad);
    if (keep == 12345.f) sink[0] = keep;
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 2000;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int nwg = prop.multiProcessorCount;
    printf("device %s, %d CUs, LDS/block %zu\n", prop.name, nwg, prop.sharedMemPerBlock);
    float *xch, *wsrc, *sink;
    unsigned *counter, *error, *mism;
    CK(hipMalloc(&xch, (size_t)2 * nwg * 64 * 4));
    CK(hipMalloc(&wsrc, (size_t)nwg * 65536));
    CK(hipMalloc(&sink, 64));
    CK(hipMalloc(&counter, 4)); CK(hipMalloc(&error, 4)); CK(hipMalloc(&mism, 4));
    CK(hipMemset(wsrc, 0, (size_t)nwg * 65536));
    const size_t lds = 140 * 1024;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(probe_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 2; ++mode) {
        for (int dma = 0; dma < 2; ++dma) {
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipMemset(counter, 0, 4)); CK(hipMemset(error, 0, 4)); CK(hipMemset(mism, 0, 4));
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(probe_kernel, dim3(nwg), dim3(1024), lds, 0, xch, counter, error, mism, rounds, mode, dma ? wsrc : nullptr, sink);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                unsigned herr, hm;
                CK(hipMemcpy(&herr, error, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hm, mism, 4, hipMemcpyDeviceToHost));
                printf("mode %d (%s) dma %d: %d rounds %.3f ms -> %.3f us / barrier round, timeout=%u mismatches=%u\n", mode, mode ? "plain + threadfence" : "agent-scope accesses",
                       dma, rounds, ms, 1000.0 * ms / rounds, herr, hm);
            }
        }
    }
    return 0;
}
