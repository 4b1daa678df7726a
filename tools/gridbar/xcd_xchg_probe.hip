// Probe 3 (round 5): what does an exchange between the 16 workgroups of ONE attention head cost when all of them sit on the SAME XCD?
//
// The fused decode-attention launch is a (16 heads, 16 sequences) grid of 1024-thread workgroups, one per CU; workgroup i runs on XCD i % 8 (tools/xcc_probe), so the
// 16 sequence-workgroups of a head share one L2.  Today each of them pulls the head's 393 KB of q/k/v weight rows through that L2 (100 MB per layer for 6.3 MB of
// weights).  The head-cooperative form needs two exchanges among those 16 workgroups:
//   A  every workgroup publishes its sequence's normalised row (4 KB)           -> barrier 1 -> every workgroup reads all 16 rows (64 KB)
//   B  every workgroup publishes its 12-column slice of q|k|v for all 16 rows   -> barrier 2 -> every workgroup reads back its own sequence's 192 values
// Round-3's grid-wide exchange (tools/gridbar/xchg_probe.hip) cost 7.9 us because flag and data cross XCDs (agent scope = memory side).  Here nothing leaves the
// XCD: plain stores (the vector L1 is write-through: an acknowledged store is in L2), the counter is bumped and polled by L2 atomics WITHOUT the sc1 bit (workgroup
// scope: executed in this XCD's L2), data is read with plain loads from lines this CU has not touched before (fresh scratch per round, like a fresh launch).
//
//   variant 0  both barriers, no data
//   variant 1  exchange A only (row publish, barrier, 64 KB read)
//   variant 2  A + B (the full protocol)
//   variant 3  like 2 with agent-scope (sc1) counter atomics: what leaving the XCD costs
// Every workgroup checks HW_REG_XCC_ID against its head-mates (published with the row) and every value it reads.  All spins are bounded.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int H = 16, B = 16, D = 1024, QKV = 192;

// POLL: 0 = workgroup-scope atomic load (the compiler emits global_load sc0: may be served by the vector L1 - can it see the other CUs' arrivals?), 1 = returning L2 atomic
// (global_atomic_add ... sc0 of 0: executed in L2 by construction), 2 = agent-scope atomic load (global_load sc1)
template <int POLL>
__device__ __forceinline__ unsigned poll(unsigned* p) {
    if (POLL == 0) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (POLL == 2) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned v, zero = 0;
    asm volatile("global_atomic_add %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p), "v"(zero) : "memory");
    return v;
}

template <int SCOPE, int POLL>
__device__ __forceinline__ bool head_barrier(unsigned* counter, unsigned target, unsigned* error) {
    __builtin_amdgcn_s_waitcnt(0);   // this wave's stores are acknowledged by L2
    __syncthreads();
    __shared__ unsigned ok_s;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, SCOPE);
        unsigned spins = 0, ok = 1;
        while (poll<POLL>(counter) < target) {
            if (++spins > (1u << 16)) { ok = 0; atomicAdd(error, 1u); break; }
            __builtin_amdgcn_s_sleep(1);
        }
        ok_s = ok;
    }
    __syncthreads();
    return ok_s != 0;
}

__device__ __forceinline__ float row_val(int r, int h, int b, int c) { return (float)((r * 31 + h * 7 + b * 3) % 101) + 0.001f * (float)(c % 500); }
__device__ __forceinline__ float qkv_val(int r, int h, int b, int j) { return (float)((r * 17 + h * 5 + b) % 89) + 0.01f * (float)j; }

template <int SCOPE, int POLL>
__global__ __launch_bounds__(1024) void xcd_xchg_kernel(float* rows, float* qkv, unsigned* xcc, unsigned* counters, unsigned* error, unsigned* mism, int rounds, int variant,
                                                        long long* t_out) {
    extern __shared__ float lds[];
    const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    unsigned xcc_id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
    xcc_id &= 0xf;
    unsigned* c1 = counters + (size_t)h * 64;        // one 256-byte slot per head
    unsigned* c2 = counters + (size_t)h * 64 + 32;
    unsigned bad = 0;
    const long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int r = 0; r < rounds; ++r) {
        float* rows_r = rows + ((size_t)r * H + h) * B * D;       // fresh lines every round
        float* qkv_r = qkv + ((size_t)r * H + h) * B * QKV;
        unsigned* xcc_r = xcc + ((size_t)r * H + h) * B;
        if (variant >= 1) {
            rows_r[(size_t)b * D + tid] = row_val(r, h, b, tid);
            if (tid == 0) xcc_r[b] = xcc_id;
        }
        if (!head_barrier<SCOPE, POLL>(c1, (unsigned)(r + 1) * B, error)) break;
        if (variant >= 1) {
            // all 16 rows: thread t reads row t >> 6, 16 floats at 16 (t & 63)
            const int rr = tid >> 6, c0 = 16 * (tid & 63);
            float4 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = *reinterpret_cast<const float4*>(rows_r + (size_t)rr * D + c0 + 4 * j);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                bad += v[j].x != row_val(r, h, rr, c0 + 4 * j) || v[j].y != row_val(r, h, rr, c0 + 4 * j + 1) || v[j].z != row_val(r, h, rr, c0 + 4 * j + 2) ||
                       v[j].w != row_val(r, h, rr, c0 + 4 * j + 3);
                lds[tid * 4 + j] = v[j].x;   // (keep the loads)
            }
            if (tid < B) bad += xcc_r[tid] != xcc_id;   // every head-mate on MY XCD?
        }
        if (variant >= 2) {
            if (tid < QKV) {   // my 12 columns for all 16 sequences
                const int bb = tid / 12, j = 12 * b + tid % 12;
                qkv_r[(size_t)bb * QKV + j] = qkv_val(r, h, bb, j);
            }
        }
        if (!head_barrier<SCOPE, POLL>(c2, (unsigned)(r + 1) * B, error)) break;
        if (variant >= 2 && tid < QKV) bad += qkv_r[(size_t)b * QKV + tid] != qkv_val(r, h, b, tid);
    }
    const long long t1 = __builtin_amdgcn_s_memrealtime();
    if (tid == 0) t_out[b * H + h] = t1 - t0;
    if (bad) atomicAdd(mism, bad);
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 200;
    setvbuf(stdout, nullptr, _IOLBF, 0);
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s, %d CUs\n", prop.name, prop.multiProcessorCount);
    if (prop.multiProcessorCount < H * B) { printf("needs %d CUs (one workgroup per CU)\n", H * B); return 1; }
    float *rows, *qkv; unsigned *xcc, *counters, *error, *mism; long long* t_out;
    CK(hipMalloc(&rows, (size_t)rounds * H * B * D * 4));
    CK(hipMalloc(&qkv, (size_t)rounds * H * B * QKV * 4));
    CK(hipMalloc(&xcc, (size_t)rounds * H * B * 4));
    CK(hipMalloc(&counters, H * 64 * 4)); CK(hipMalloc(&error, 4)); CK(hipMalloc(&mism, 4)); CK(hipMalloc(&t_out, H * B * 8));
    const size_t lds = 140 * 1024;   // one workgroup per CU, like the decode kernel with its staged K/V
#define WG __HIP_MEMORY_SCOPE_WORKGROUP
#define AG __HIP_MEMORY_SCOPE_AGENT
    typedef void (*Kern)(float*, float*, unsigned*, unsigned*, unsigned*, unsigned*, int, int, long long*);
    struct Var { const char* name; Kern k; int data; };
    const Var vars[] = {
        {"two head barriers, no data; poll = returning L2 atomic", xcd_xchg_kernel<WG, 1>, 0},
        {"A only (4 KB row publish, barrier, 64 KB read, empty barrier 2); poll = returning L2 atomic", xcd_xchg_kernel<WG, 1>, 1},
        {"A + B, the full protocol; poll = returning L2 atomic", xcd_xchg_kernel<WG, 1>, 2},
        {"A + B; poll = workgroup-scope load (global_load sc0)", xcd_xchg_kernel<WG, 0>, 2},
        {"A + B; arrive workgroup scope, poll = agent-scope load (global_load sc1)", xcd_xchg_kernel<WG, 2>, 2},
        {"A + B; arrive AND poll agent scope (sc1 atomics: the memory side)", xcd_xchg_kernel<AG, 2>, 2},
    };
    for (const Var& v : vars) CK(hipFuncSetAttribute(reinterpret_cast<const void*>(v.k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int v = 0; v < (int)(sizeof(vars) / sizeof(vars[0])); ++v) {
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemset(counters, 0, H * 64 * 4)); CK(hipMemset(error, 0, 4)); CK(hipMemset(mism, 0, 4));
            CK(hipMemset(rows, 0, (size_t)rounds * H * B * D * 4)); CK(hipMemset(qkv, 0, (size_t)rounds * H * B * QKV * 4));
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(vars[v].k, dim3(H, B), dim3(1024), lds, 0, rows, qkv, xcc, counters, error, mism, rounds, vars[v].data, t_out);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            unsigned herr, hm;
            long long ht[H * B];
            CK(hipMemcpy(&herr, error, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hm, mism, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(ht, t_out, sizeof(ht), hipMemcpyDeviceToHost));
            long long tmax = 0, tmin = 1LL << 60;
            for (int i = 0; i < H * B; ++i) { tmax = ht[i] > tmax ? ht[i] : tmax; tmin = ht[i] < tmin ? ht[i] : tmin; }
            printf("variant %d (%s): %.3f us / round by events; in-kernel %.3f .. %.3f us / round (100 MHz clock); timeouts=%u mismatches=%u\n", v, vars[v].name, 1000.0 * ms / rounds,
                   tmin / 100.0 / rounds, tmax / 100.0 / rounds, herr, hm);
        }
    }
    return 0;
}
