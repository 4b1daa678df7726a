// Probe 2 for the persistent decode-step design: cost of ONE all-to-all activation exchange between phases on MI355X (8 XCDs, one L2 each).
// Round r: every workgroup writes its 16 x 16 block of X[16][4096] (the shape of an MLP-up output), then every workgroup reads a 16 x 1024 slice (64 KB).
//   variant 0  barrier only (agent-scope counter, no data)
//   variant 1  sc1 stores -> counter barrier -> sc1 dword loads (every load goes to the memory side)
//   variant 2  sc1 stores -> counter barrier -> ONE wave per workgroup executes an agent acquire fence (buffer_inv sc1) -> plain 16-byte loads (L2 can serve 31 of 32)
//   variant 3  (value, epoch) packed in 64-bit sc1 stores; consumers spin on the data itself (no barrier)
//   variant 4  like 2, but the fence is executed by every wave
// All spins are bounded.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
#define AG __HIP_MEMORY_SCOPE_AGENT

struct Bar { unsigned* counter; unsigned* error; unsigned nwg; unsigned gen; };

__device__ __forceinline__ void bar_sync(Bar& b) {
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    b.gen += 1;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(b.counter, 1u, __ATOMIC_RELAXED, AG);
        const unsigned target = b.gen * b.nwg;
        unsigned spins = 0;
        while (__hip_atomic_load(b.counter, __ATOMIC_RELAXED, AG) < target) {
            if (++spins > (1u << 16) || __hip_atomic_load(b.error, __ATOMIC_RELAXED, AG)) { __hip_atomic_store(b.error, 1u, __ATOMIC_RELAXED, AG); break; }
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(1024) void xchg_kernel(float* X, unsigned long long* X2, unsigned* counter, unsigned* error, unsigned* mism, int rounds, int variant, long long* tstamp) {
    extern __shared__ float lds[];
    Bar b{counter, error, gridDim.x, 0};
    const int wg = blockIdx.x, tid = threadIdx.x, nwg = gridDim.x, wave = tid >> 6;
    const int ncols = nwg * 16;   // 4096 at 256 workgroups
    unsigned bad = 0;
    for (int r = 0; r < rounds; ++r) {
        if (__hip_atomic_load(error, __ATOMIC_RELAXED, AG)) break;   // a timed-out barrier: drain
        float* Xr = X + (size_t)(r & 1) * 16 * ncols;
        unsigned long long* X2r = X2 + (size_t)(r & 1) * 16 * ncols;
        // ---- produce: 16 x 16 block, thread t < 256 -> (row t >> 4, col 16 wg + (t & 15))
        if (tid < 256) {
            const int row = tid >> 4, col = wg * 16 + (tid & 15);
            const float v = (float)(r % 97) + 0.5f * row + 0.001f * (col % 512);
            if (variant == 3) {
                const unsigned long long pk = ((unsigned long long)(unsigned)(r + 1) << 32) | __float_as_uint(v);
                __hip_atomic_store(X2r + (size_t)row * ncols + col, pk, __ATOMIC_RELAXED, AG);
            } else if (variant != 0) {
                __hip_atomic_store(Xr + (size_t)row * ncols + col, v, __ATOMIC_RELAXED, AG);
            }
        }
        if (variant != 3) bar_sync(b);
        // ---- consume: 16 rows x 1024 cols starting at col0 = (wg % 4) * 1024 (mod ncols); thread t -> row t >> 6, cols 16 (t & 63) .. + 15
        const int row = tid >> 6, c0 = ((wg & 3) * 1024) % ncols + 16 * (tid & 63);
        float got[16];
        if (variant == 1) {
#pragma unroll
            for (int j = 0; j < 16; ++j) got[j] = __hip_atomic_load(Xr + (size_t)row * ncols + c0 + j, __ATOMIC_RELAXED, AG);
        } else if (variant == 2 || variant == 4) {
            if (variant == 4 || wave == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            if (variant == 2) __syncthreads();
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 q = *reinterpret_cast<const float4*>(Xr + (size_t)row * ncols + c0 + 4 * j);
                got[4 * j] = q.x; got[4 * j + 1] = q.y; got[4 * j + 2] = q.z; got[4 * j + 3] = q.w;
            }
        } else if (variant == 3) {
            unsigned spins = 0;
            bool ok = false;
            while (!ok) {
                ok = true;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const unsigned long long pk = __hip_atomic_load(X2r + (size_t)row * ncols + c0 + j, __ATOMIC_RELAXED, AG);
                    got[j] = __uint_as_float((unsigned)pk);
                    ok = ok && (unsigned)(pk >> 32) == (unsigned)(r + 1);
                }
                if (++spins > (1u << 12) || __hip_atomic_load(error, __ATOMIC_RELAXED, AG)) { __hip_atomic_store(error, 1u, __ATOMIC_RELAXED, AG); break; }
            }
            // a consumer must not run two rounds ahead of a producer that still has to read the buffer being overwritten: rounds alternate buffers and
            // every workgroup both produces and consumes each round, so a producer of round r+2 has consumed round r+1, which needs everyone's round r+1 block,
            // which they write after consuming round r.  Safe with two buffers.
        }
        if (variant != 0) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const float want = (float)(r % 97) + 0.5f * row + 0.001f * ((c0 + j) % 512);
                if (got[j] != want) bad += 1;
            }
        }
        if (variant == 2 || variant == 4) __syncthreads();
    }
    if (bad) atomicAdd(mism, bad);
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 1000;
    setvbuf(stdout, nullptr, _IOLBF, 0);
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int nwg = prop.multiProcessorCount;
    float* X; unsigned long long* X2; unsigned *counter, *error, *mism;
    CK(hipMalloc(&X, (size_t)2 * 16 * nwg * 16 * 4));
    CK(hipMalloc(&X2, (size_t)2 * 16 * nwg * 16 * 8));
    CK(hipMalloc(&counter, 4)); CK(hipMalloc(&error, 4)); CK(hipMalloc(&mism, 4));
    const size_t lds = 140 * 1024;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(xchg_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char* names[] = {"barrier only", "sc1 stores, barrier, sc1 dword loads", "sc1 stores, barrier, one-wave acquire fence, plain loads", "(value, epoch) 64-bit sc1, spin on data",
                           "sc1 stores, barrier, every-wave acquire fence, plain loads"};
    for (int v = 0; v < 5; ++v) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipMemset(counter, 0, 4)); CK(hipMemset(error, 0, 4)); CK(hipMemset(mism, 0, 4));
            CK(hipMemset(X2, 0, (size_t)2 * 16 * nwg * 16 * 8));
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(xchg_kernel, dim3(nwg), dim3(1024), lds, 0, X, X2, counter, error, mism, rounds, v, nullptr);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            unsigned herr, hm;
            CK(hipMemcpy(&herr, error, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hm, mism, 4, hipMemcpyDeviceToHost));
            printf("variant %d (%s): %.3f us / round, timeout=%u mismatches=%u\n", v, names[v], 1000.0 * ms / rounds, herr, hm);
            fflush(stdout);
        }
    }
    return 0;
}
