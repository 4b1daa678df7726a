// Route A decode step, fused form: three launches per transformer layer instead of eight.
//
// A decode step pushes ONE new row per sequence through the 24 layers (Block.forward, transformer/mingpt_sparse.py:240-253:
// x = ln1(x); x = x + attn(x); x = x + mlp(ln2(x))).  With 16-64 rows the step is a chain of short, strictly dependent kernels, and what
// it costs is the fixed price of every link (launch + ramp + tail, 4-7 us each on MI355X), not arithmetic.  The row only has three
// global dependencies per layer (the attention of a head needs that head's q/k/v; ln2 needs the whole attention row; the MLP
// down-projection needs the whole hidden row), so the layer is cut exactly there:
//
//   ar_attn_fused_kernel   one workgroup per (layout group, head):  reduce the previous layer's split-K partials (+bias +residual)
//                          -> ln1 -> q/k/v projection of THIS head (192 rows of the fused QKV weight, shared through the XCD's L2 by the
//                          sequences of the head) -> append k/v to the cache -> softmax(dh^-0.5 (q k^T + camera bias) + mask) v over the
//                          cache (SparseSelfAttention.forward, transformer/sparse_self_attention.py:150-176) -> + ln1(x) -> x2
//   skinny_fused_kernel<LN>   ln2 fused into the MLP up-projection (+bias, GELU): every workgroup normalises the <= 16 rows itself
//                          while its weight slice is in flight
//   skinny_fused_kernel<no LN> MLP down-projection, split over K across workgroups; the partial sums are NOT reduced by a kernel of
//                          their own: the consumer (next layer's ar_attn_fused_kernel / the head) adds them in a fixed order.
//
// Shared condition prefix (BASELINE config 5, several samples per BEV layout): a workgroup serves the G sequences of one layout
// and one head; the K prefix rows are streamed ONCE and scored against the G queries, the private suffixes are split over wave teams.
#include <atomic>
#include <hip/hip_ext.h>
#include <type_traits>

#include "common.h"
#include "kernels.h"
#include "profiler.h"

namespace bevgen {

// ----------------------------------------------------------------------------------------------------------------- row source
// Every load of an element is requested before the first one is used, and UNCONDITIONALLY: a predicated load (`k < ns ? p[k] : 0`) becomes a
// divergent branch whose join point waits for the load, i.e. one serialised memory round trip per partial.  The launchers therefore make every
// pointer valid (rowsrc_fix: absent partials / bias alias `base`) and the kernel only selects which values count.
constexpr int ROWSRC_MAX_SPLITS = 8;   // (4 K slices of the MLP-down launch; 8 XCD planes of the fused MLP launch)
constexpr int ROWSRC_RS_SPLITS = 4;    // the row-source form of skinny_fused_kernel (decode_path = split) keeps its register budget: at most 4 partials
inline RowSrc rowsrc_fix(RowSrc r) {
    if (!r.partial || r.ns == 0) { r.partial = r.base; r.ns = 0; r.pstride = 0; r.pld = r.ld; }
    r.has_bias = r.bias != nullptr;
    if (!r.bias) r.bias = r.base;
    return r;
}
__device__ __forceinline__ float rowsrc_at(const RowSrc& r, int m, int c) {
    float p[ROWSRC_MAX_SPLITS];
#pragma unroll
    for (int k = 0; k < ROWSRC_MAX_SPLITS; ++k) p[k] = r.partial[(long)(k < r.ns ? k : 0) * r.pstride + (long)m * r.pld + c];
    const float b = r.bias[r.has_bias ? c : 0];
    const float x = r.base[(long)m * r.ld + c];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < ROWSRC_MAX_SPLITS; ++k) s += k < r.ns ? p[k] : 0.f;
    return (s + (r.has_bias ? b : 0.f)) + x;
}
__global__ __launch_bounds__(256) void rowsrc_materialize_kernel(RowSrc r, float* __restrict__ out, int M, int D, int* counter) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long ic = i < (long)M * D ? i : 0;
    const float v = rowsrc_at(r, (int)(ic / D), (int)(ic % D));
    if (i < (long)M * D) out[i] = v;
    if (counter && i == 0) *counter += 1;   // the step counter: no kernel of this launch reads it
}

void launch_rowsrc_materialize(const RowSrc& r0, float* out, int M, int D, int* counter, hipStream_t s) {
    const RowSrc r = rowsrc_fix(r0);
    hipLaunchKernelGGL(rowsrc_materialize_kernel, dim3(cdiv((long)M * D, 256)), dim3(256), 0, s, r, out, M, D, counter);
    LAUNCH_CHECK();
}

// ----------------------------------------------------------------------------------------------------------------- helpers
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
// K / V rows are read with the NON-TEMPORAL hint (registers and LDS-DMA alike): a decode step streams ~85 MB of them per layer through 8 x 4 MB of L2 exactly once, and
// without the hint that stream evicts what the step does reuse - the head's q/k/v weight rows its 16 workgroups share, and the weight images the MLP launches park in L2
// for each other.  Same-box A/B, fp16 cache + fp16 weights: 0.996-0.999 -> 0.959-0.964 ms/step (profiles/r04_ab_kv_nontemporal.txt).
template <int DT> struct KvRow;
template <> struct KvRow<0> {  // fp32 rows: 256 B = 16 lanes x 16 B
    static constexpr int LPK = 16, DPL = 4;
    typedef f32x4 Raw;
    __device__ static __forceinline__ Raw load(const void* base, long row, int sub) {
        return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(base) + row * 64 + sub * 4));
    }
    __device__ static __forceinline__ float get(const Raw& r, int i) { return r[i]; }
};
template <> struct KvRow<1> {  // fp16 rows: 128 B = 8 lanes x 16 B; kept packed in registers, widened at the point of use
    static constexpr int LPK = 8, DPL = 8;
    typedef half8_t Raw;
    __device__ static __forceinline__ Raw load(const void* base, long row, int sub) {
        return __builtin_nontemporal_load(reinterpret_cast<const half8_t*>(reinterpret_cast<const _Float16*>(base) + row * 64 + sub * 8));
    }
    __device__ static __forceinline__ float get(const Raw& r, int i) { return (float)r[i]; }
};

// sum over aligned groups of 8 / 16 lanes with DPP row operations (every lane of the group ends with the group's sum)
template <int LPK>
__device__ __forceinline__ float lane_group_sum(float d) {
    d += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(d), 0xB1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
    d += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(d), 0x4E, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
    d += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(d), 0x141, 0xf, 0xf, true));   // row_half_mirror
    if (LPK == 16) d += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(d), 0x140, 0xf, 0xf, true));   // row_mirror
    return d;
}
__device__ __forceinline__ float wave_sum_dpp(float d) {
    d = lane_group_sum<16>(d);
    d += xor16(d);
    return d + xor32(d);
}
// Wave sums of FOUR per-lane values for the price of one and a half: the two cross-half / cross-row steps each fold two values into one register (fold32 / fold16),
// the four in-row steps then run once.  Lane L ends with the total of v[2 bit5(L) + bit4(L)] (every lane of a 16-lane row holds it).
__device__ __forceinline__ float wave_sum4(float v0, float v1, float v2, float v3) {
    return lane_group_sum<16>(fold16(fold32(v0, v2), fold32(v1, v3)));
}
// two values: lane L ends with the total of v[bit5(L)]
__device__ __forceinline__ float wave_sum2(float v0, float v1) {
    float t = fold32(v0, v1);
    t += xor16(t);
    return lane_group_sum<16>(t);
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt: every global load / LDS-DMA a wave has in flight must land before it passes - in
// the fused decode kernel that makes the LayerNorm barriers wait for the K/V pieces and weight rows streaming behind them (statistics done at 6.5 instead of 3.9 us).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ float4 mul4(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
constexpr long long MLPF_TIMEOUT_TICKS = 200LL * 1000 * 100;   // bound of the XCD-local barrier spin (ar_mlp_fused_kernel): 200 ms of the 100 MHz clock
constexpr float kLog2eF = 1.44269504088896340736f;
constexpr int AF_WAVES = 16;

// One online-softmax state per query: running maximum (base-2 domain), denominator, weighted value sum of this lane's DPL dims.
template <int DT, int NQ, int U>
struct Attend {
    using T = KvRow<DT>;
    static constexpr int LPK = T::LPK, DPL = T::DPL;
    struct Buf { typename T::Raw k[U], v[U]; };

    // A wave walks 16-key chunks.  KPS keys per pipeline step (U loads of KPI consecutive keys = 1 KiB each), SPC steps per chunk.
    static constexpr int KPI = 64 / LPK, KPS = U * KPI, SPC = 16 / KPS;
    static_assert(KPS <= 16 && 16 % KPS == 0, "a pipeline step must tile a 16-key chunk");

    template <int USTRIDE>
    __device__ static __forceinline__ void load(Buf& b, const void* kc, const void* vc, long row0, int key0, int k_end, int sub) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int kcl = min(key0 + u * USTRIDE, k_end - 1);   // clamped address, masked in compute()
            b.k[u] = T::load(kc, row0 + kcl, sub);
            b.v[u] = T::load(vc, row0 + kcl, sub);
        }
    }
    template <int USTRIDE>
    __device__ static __forceinline__ void compute(const Buf& b, int key0, int k_end, const float* bias_s, const float (&q)[NQ][DPL], float (&m)[NQ],
                                                   float (&l)[NQ], float (&acc)[NQ][DPL]) {
        float sc[NQ][U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int key = key0 + u * USTRIDE;
            const float bv = key < k_end ? bias_s[min(key, k_end - 1)] : kNegBig;
#pragma unroll
            for (int qi = 0; qi < NQ; ++qi) {
                float d = 0.f;
#pragma unroll
                for (int i = 0; i < DPL; ++i) d = fmaf(q[qi][i], T::get(b.k[u], i), d);
                d = lane_group_sum<LPK>(d);
                sc[qi][u] = bv <= kNegBig ? kNegBig : d + bv;
            }
        }
#pragma unroll
        for (int qi = 0; qi < NQ; ++qi) {
            float mx = m[qi];
#pragma unroll
            for (int u = 0; u < U; ++u) mx = fmaxf(mx, sc[qi][u]);
            const float alpha = __builtin_amdgcn_exp2f(m[qi] - mx);
            l[qi] *= alpha;
#pragma unroll
            for (int i = 0; i < DPL; ++i) acc[qi][i] *= alpha;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float p = sc[qi][u] <= kNegBig ? 0.f : __builtin_amdgcn_exp2f(sc[qi][u] - mx);
                l[qi] += p;
#pragma unroll
                for (int i = 0; i < DPL; ++i) acc[qi][i] = fmaf(p, T::get(b.v[u], i), acc[qi][i]);
            }
            m[qi] = mx;
        }
    }
    // The walk covers list positions [lo, hi).  With a chunk list (block-sparse layouts: only chunks that hold a present block are listed) position p stands for
    // chunk list_s[p]: wave `slot` of a team of `nslots` takes positions lo + slot, lo + slot + nslots, ...; step j of its walk reads keys
    // 16 c + (j % SPC) KPS + u KPI + kslot - KPI consecutive keys = 1 KiB per load.  Without a list (every block at or below the diagonal present) the range
    // [16 lo, min(16 hi, k_end)) is walked in the interleaved order the dense kernel always used: at step j, load u of wave `slot` covers keys
    // 16 lo + (j U + u) nslots KPI + slot KPI + kslot, so that the team's waves read ONE contiguous nslots KiB per load instruction (measured: the per-wave chunk
    // order costs 5 % of the K/V rate at density 1).  Two register buffers: the loads of step j + 1 are in flight while step j is scored.
    // (LIST and the team size are compile-time: every stride of the walk is then a constant, as in the dense kernel of round 2 - with run-time strides the
    // address arithmetic of the eight loads per step cost 1.8 us of the 27 us K/V phase)
    template <bool LIST, int NSLOTS>
    __device__ static __forceinline__ void run_as(const void* kc, const void* vc, long row0, const uint16_t* list_s, int lo, int hi, int slot, int kslot, int k_end,
                                                  int sub, const float* bias_s, const float (&q)[NQ][DPL], float (&m)[NQ], float (&l)[NQ], float (&acc)[NQ][DPL]) {
        constexpr int USTRIDE = LIST ? KPI : NSLOTS * KPI, SPAN = U * NSLOTS * KPI;
        int steps;
        if (LIST) {
            const int mine = hi - lo - slot;
            steps = mine > 0 ? ((mine + NSLOTS - 1) / NSLOTS) * SPC : 0;   // wave-uniform
        } else {
            const int len = min(16 * hi, k_end) - 16 * lo;
            steps = len > 0 ? (len + SPAN - 1) / SPAN : 0;
        }
        if (steps == 0) return;
        auto key_of = [&](int j) {
            if (LIST) return 16 * (int)list_s[lo + slot + (j / SPC) * NSLOTS] + (j % SPC) * KPS + kslot;
            return 16 * lo + j * SPAN + slot * KPI + kslot;
        };
        Buf b0, b1;
        load<USTRIDE>(b0, kc, vc, row0, key_of(0), k_end, sub);
        for (int j = 0; j < steps; j += 2) {
            if (j + 1 < steps) load<USTRIDE>(b1, kc, vc, row0, key_of(j + 1), k_end, sub);
            compute<USTRIDE>(b0, key_of(j), k_end, bias_s, q, m, l, acc);
            if (j + 2 < steps) load<USTRIDE>(b0, kc, vc, row0, key_of(j + 2), k_end, sub);
            if (j + 1 < steps) compute<USTRIDE>(b1, key_of(j + 1), k_end, bias_s, q, m, l, acc);
        }
    }
    // Dense walk over the keys [0, k_end) by a team of NSLOTS waves whose leading `js` pipeline steps were STAGED in LDS while the kernel's prologue ran:
    // step j of wave `slot` = U K pieces + U V pieces of 1 KiB (piece p = j U + u = keys p NSLOTS KPI + slot KPI .. + KPI - 1), staged step-major at
    // st + (2 U j + u) KiB (K) / st + (2 U j + U + u) KiB (V), st = the wave's region + lane * 16.  The pieces were requested by LDS-DMA earlier; the walk opens by requesting
    // its first TWO global steps - they queue behind the pieces in the CU's in-order memory pipeline and keep HBM busy - then waits until only those are outstanding (every
    // piece has landed: loads return in issue order), scores the staged steps out of LDS, and continues with the global steps already in flight: no cold start.
    // LIST: the same for the chunk-list walk (block-sparse layouts): step j of wave `slot` = the 16 keys of chunk list_s[slot + j NSLOTS] (one chunk per step: SPC == 1),
    // positions [0, hi); the staged steps are this wave's leading WHOLE chunks below the new key.
    template <int NSLOTS, bool LIST = false>
    __device__ static __forceinline__ void run_staged(const void* kc, const void* vc, long row0, int slot, int kslot, int k_end, int sub, const float* bias_s, const char* st, int js,
                                                      const float (&q)[NQ][DPL], float (&m)[NQ], float (&l)[NQ], float (&acc)[NQ][DPL], const uint16_t* list_s = nullptr,
                                                      int hi = 0) {
        static_assert(!LIST || SPC == 1, "the staged chunk-list walk takes one chunk per pipeline step");
        constexpr int USTRIDE = LIST ? KPI : NSLOTS * KPI, SPAN = U * NSLOTS * KPI;
        int steps;
        if (LIST) steps = hi - slot > 0 ? (hi - slot + NSLOTS - 1) / NSLOTS : 0;
        else steps = k_end > 0 ? (k_end + SPAN - 1) / SPAN : 0;   // js <= steps: staged steps are whole steps below k_end
        auto key_of = [&](int j) {
            if (LIST) return 16 * (int)list_s[slot + j * NSLOTS] + kslot;
            return j * SPAN + slot * KPI + kslot;
        };
        Buf b0, b1;
        if (js + 1 < steps) {
            load<USTRIDE>(b0, kc, vc, row0, key_of(js), k_end, sub);
            load<USTRIDE>(b1, kc, vc, row0, key_of(js + 1), k_end, sub);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * U) : "memory");
        } else if (js < steps) {
            load<USTRIDE>(b0, kc, vc, row0, key_of(js), k_end, sub);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * U) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        for (int j = 0; j < js; ++j) {   // (one K/V piece pair at a time: a third U-deep register buffer next to b0 / b1 spills in the fp32-cache variants)
#pragma unroll
            for (int u = 0; u < U; ++u) {
                typename Attend<DT, NQ, 1>::Buf t;
                t.k[0] = *reinterpret_cast<const typename T::Raw*>(st + (2 * U * j + u) * 1024);
                t.v[0] = *reinterpret_cast<const typename T::Raw*>(st + (2 * U * j + U + u) * 1024);
                Attend<DT, NQ, 1>::template compute<USTRIDE>(t, key_of(j) + u * USTRIDE, k_end, bias_s, q, m, l, acc);
            }
        }
        for (int j = js; j < steps; j += 2) {
            compute<USTRIDE>(b0, key_of(j), k_end, bias_s, q, m, l, acc);
            if (j + 2 < steps) load<USTRIDE>(b0, kc, vc, row0, key_of(j + 2), k_end, sub);
            if (j + 1 < steps) {
                compute<USTRIDE>(b1, key_of(j + 1), k_end, bias_s, q, m, l, acc);
                if (j + 3 < steps) load<USTRIDE>(b1, kc, vc, row0, key_of(j + 3), k_end, sub);
            }
        }
    }
};

// merge the key slots of a wave (lanes with equal `sub`); afterwards every lane holds the wave's state for its dims
template <int LPK, int DPL>
__device__ __forceinline__ void wave_merge(float& m, float& l, float (&acc)[DPL]) {
    float m_all = m;
#pragma unroll
    for (int o = LPK; o < 64; o <<= 1) m_all = fmaxf(m_all, __shfl_xor(m_all, o, 64));
    const float f = __builtin_amdgcn_exp2f(m - m_all);
    l *= f;
#pragma unroll
    for (int i = 0; i < DPL; ++i) acc[i] *= f;
#pragma unroll
    for (int o = LPK; o < 64; o <<= 1) {
        l += __shfl_xor(l, o, 64);
#pragma unroll
        for (int i = 0; i < DPL; ++i) acc[i] += __shfl_xor(acc[i], o, 64);
    }
    m = m_all;
}

// key loads per lane group and pipeline step of the private walk (fp16 rows: 3 or 4 measured no faster)
template <int DT, int G> struct WalkU { static constexpr int value = (G == 1 && DT == 0) ? 4 : 2; };

#define AF_TRACE(i) do { if (a.trace && tid == 0) a.trace[(long)(blockIdx.y * gridDim.x + blockIdx.x) * 8 + (i)] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)

// Second half of both attention kernels: this step's k / v rows (qkv_s, LDS) go into the cache, the walk over the visible 16-key chunks, the merge of the
// 16 waves, + residual (res_s[g * ldres + d], LDS) -> out.  Called by every thread of the workgroup after a barrier that made qkv_s / bias_s / the list visible.
// SP: the walk follows a chunk list (block-sparse layout); false = the dense interleaved walk with every stride a constant.
// STG (G = 1): the walk opens with `js` steps staged in LDS at `st` (Attend::run_staged)
template <int DT, int G, bool SP, bool STG = false>
__device__ __forceinline__ void af_append_attend_store(const ArAttnFusedArgs& a, const float* bias_s, const float* qkv_s, float* red, const uint16_t* walk, int n_pos, int p_pos,
                                                       int n, int head, int b0, const float* res_s, int ldres, const char* st = nullptr, int js = 0) {
    static_assert(!STG || G == 1, "staged steps are defined on the walk of one sequence");
    using T = KvRow<DT>;
    constexpr int LPK = T::LPK, DPL = T::DPL, NW = AF_WAVES, TW = NW / G;
    constexpr int U = WalkU<DT, G>::value;
    constexpr int UP = G >= 4 ? 1 : U;   // same for the shared-prefix phase (G queries' state lives in registers)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sub = lane % LPK, kslot = lane / LPK;
    const int row = n - 1;
    const float sl2 = a.scale * kLog2eF;
    // ---- append this step's k / v rows to the cache (row n-1 of every sequence of the group).  The row is NOT read back by this launch: the walk below covers the keys
    //      [0, n-1) that were in the cache when the kernel started, and the new key joins in the merge straight from LDS (rounded to the cache's storage type, so that
    //      it counts exactly as the next step will read it).  No barrier, no wait for the store: nothing in this launch depends on it.
    //      (key split: the workgroup that owns the LAST list positions appends and counts the new key)
    if (tid < 128 * G && blockIdx.z + 1 == gridDim.z) {
        const int g = tid >> 7, is_v = (tid >> 6) & 1, d = tid & 63;
        const float val = qkv_s[g * 192 + 64 + is_v * 64 + d];
        void* cache = is_v ? a.vcache : a.kcache;
        const long idx = ((((long)(b0 + g)) * a.H + head) * a.Lmax + row) * 64 + d;
        if (DT == 0) reinterpret_cast<float*>(cache)[idx] = val;
        else {
            const _Float16 hv = (_Float16)val;
            unsigned bad = 0;
            guard_half(hv, bad);   // a key / value outside the fp16 range of the cache (or NaN)
            if (bad) status_raise(a.status, BG_ST_F16_RANGE);
            reinterpret_cast<_Float16*>(cache)[idx] = hv;
        }
    }
    const int n_old = n - 1;   // keys the walk covers

    AF_TRACE(3);
    // ---- attention
    float* my_red = red + wave * (G + 1) * 66;
    if (G == 1) {
        float q[1][DPL], m[1] = {kNegBig}, l[1] = {0.f}, acc[1][DPL];
#pragma unroll
        for (int i = 0; i < DPL; ++i) { q[0][i] = qkv_s[sub * DPL + i] * sl2; acc[0][i] = 0.f; }
        const long row0 = ((long)b0 * a.H + head) * a.Lmax;
        // (key split: this workgroup's share of the list positions; gridDim.z == 1 -> all of them)
        const int ksl = (int)gridDim.z, kz = (int)blockIdx.z;
        const int p_lo = (int)((long)kz * n_pos / ksl), p_hi = (int)((long)(kz + 1) * n_pos / ksl);
        // (the dense walk strides over whole 256-key spans: its end must be clipped to this share's last key, or the next share's keys are counted twice)
        const int k_hi = SP ? n_old : min(n_old, 16 * p_hi);
        if constexpr (STG) Attend<DT, 1, U>::template run_staged<NW, SP>(a.kcache, a.vcache, row0, wave, kslot, k_hi, sub, bias_s, st, js, q, m, l, acc, walk, p_hi);
        else Attend<DT, 1, U>::template run_as<SP, NW>(a.kcache, a.vcache, row0, walk, p_lo, p_hi, wave, kslot, k_hi, sub, bias_s, q, m, l, acc);
        wave_merge<LPK, DPL>(m[0], l[0], acc[0]);
        if (kslot == 0) {
            if (sub == 0) { my_red[0] = m[0]; my_red[1] = l[0]; }
#pragma unroll
            for (int i = 0; i < DPL; ++i) my_red[2 + sub * DPL + i] = acc[0][i];
        }
    } else {
        // shared prefix [0, prefix): rows of the group's first sequence, every key row scored against the G queries
        {
            float q[G][DPL], m[G], l[G], acc[G][DPL];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                m[g] = kNegBig; l[g] = 0.f;
#pragma unroll
                for (int i = 0; i < DPL; ++i) { q[g][i] = qkv_s[g * 192 + sub * DPL + i] * sl2; acc[g][i] = 0.f; }
            }
            const long row0 = ((long)b0 * a.H + head) * a.Lmax;
            Attend<DT, G, UP>::template run_as<SP, NW>(a.kcache, a.vcache, row0, walk, 0, p_pos, wave, kslot, min(a.prefix, n_old), sub, bias_s, q, m, l, acc);
#pragma unroll
            for (int g = 0; g < G; ++g) {
                wave_merge<LPK, DPL>(m[g], l[g], acc[g]);
                if (kslot == 0) {
                    if (sub == 0) { my_red[g * 66] = m[g]; my_red[g * 66 + 1] = l[g]; }
#pragma unroll
                    for (int i = 0; i < DPL; ++i) my_red[g * 66 + 2 + sub * DPL + i] = acc[g][i];
                }
            }
        }
        // private suffix [prefix, n): team of TW waves per sequence
        {
            const int g_own = wave / TW, wt = wave % TW;
            float q[1][DPL], m[1] = {kNegBig}, l[1] = {0.f}, acc[1][DPL];
#pragma unroll
            for (int i = 0; i < DPL; ++i) { q[0][i] = qkv_s[g_own * 192 + sub * DPL + i] * sl2; acc[0][i] = 0.f; }
            const long row0 = ((long)(b0 + g_own) * a.H + head) * a.Lmax;
            Attend<DT, 1, U>::template run_as<SP, TW>(a.kcache, a.vcache, row0, walk, p_pos, n_pos, wt, kslot, n_old, sub, bias_s, q, m, l, acc);
            wave_merge<LPK, DPL>(m[0], l[0], acc[0]);
            if (kslot == 0) {
                if (sub == 0) { my_red[G * 66] = m[0]; my_red[G * 66 + 1] = l[0]; }
#pragma unroll
                for (int i = 0; i < DPL; ++i) my_red[G * 66 + 2 + sub * DPL + i] = acc[0][i];
            }
        }
    }
    __syncthreads();
    AF_TRACE(4);

    // ---- merge the waves (fixed order), normalise, add the ln1(x) residual (Block.forward: the residual is the NORMALISED row)
    if (tid < 64 * G) {
        const int g = tid >> 6, d = tid & 63;
        float mm = kNegBig;
#pragma unroll
        for (int w = 0; w < NW; ++w) mm = fmaxf(mm, red[(w * (G + 1) + g) * 66]);
        if (G > 1) {
#pragma unroll
            for (int w = 0; w < TW; ++w) mm = fmaxf(mm, red[((g * TW + w) * (G + 1) + G) * 66]);
        }
        float l = 0.f, o = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const float* e = red + (w * (G + 1) + g) * 66;
            const float f = __builtin_amdgcn_exp2f(e[0] - mm);
            l += e[1] * f;
            o += e[2 + d] * f;
        }
        if (G > 1) {
#pragma unroll
            for (int w = 0; w < TW; ++w) {
                const float* e = red + ((g * TW + w) * (G + 1) + G) * 66;
                const float f = __builtin_amdgcn_exp2f(e[0] - mm);
                l += e[1] * f;
                o += e[2 + d] * f;
            }
        }
        if (blockIdx.z + 1 == gridDim.z) {   // the new key (row n-1), from LDS: wave g = sequence g, lane d = dim d
            float kn = qkv_s[g * 192 + 64 + d], vn = qkv_s[g * 192 + 128 + d];
            if (DT == 1) { kn = (float)(_Float16)kn; vn = (float)(_Float16)vn; }
            const float bv = bias_s[row];
            const float sn = wave_sum_dpp(qkv_s[g * 192 + d] * sl2 * kn) + bv;
            if (bv > kNegBig) {
                const float m2 = fmaxf(mm, sn), f = __builtin_amdgcn_exp2f(mm - m2), pn = __builtin_amdgcn_exp2f(sn - m2);
                l = l * f + pn;
                o = o * f + pn * vn;
                mm = m2;
            }
        }
        if (gridDim.z > 1) {   // key split (G == 1): partial state for the combine kernel
            float* pw = a.kws + ((((long)b0 * a.H + head) * gridDim.z) + blockIdx.z) * 66;
            if (d == 0) { pw[0] = mm; pw[1] = l; }
            pw[2 + d] = o;
        } else {
            a.out[(long)(b0 + g) * a.ldo + head * 64 + d] = o / l + res_s[g * ldres + d];
        }
    }
    AF_TRACE(5);
}

// Tail of the fused kernel: block layout (kernel-uniform branches: keys of absent blocks are hidden, the walk follows the list of chunks that
// hold a present block; list positions [0, n_pos) are the chunks that start below n, ascending; without a list position = chunk), then append + walk + merge + store.
template <int DT, int G, bool SP, bool STG>
__device__ __forceinline__ void af_layout_attend(const ArAttnFusedArgs& a, const SparseVis& vis, const uint8_t* lay_row, const uint16_t* chunk_row, float* bias_s, const float* qkv_s,
                                                 float* red, uint16_t* list_s, int n, int head, int b0, const float* res_s, int ldres, const char* st, int js) {
    const int tid = threadIdx.x;
    int n_pos, p_pos;
    if (SP) {   // (the dense instantiation carries none of this: the launcher picks SP only when a layout hides something)
        if (vis.has_lay)
            for (int k = tid; k < n; k += 1024)
                if (!lay_row[k / vis.blk]) bias_s[k] = kNegBig;
        const int chunk_total = chunk_row[0];
        const int chunk_id = chunk_row[min(1 + tid, vis.chunks_ld - 1)];
        if (tid < chunk_total) list_s[tid] = (uint16_t)chunk_id;
        n_pos = __syncthreads_count(tid < chunk_total && chunk_id * 16 < n);
        // shared prefix (G > 1): positions [0, p_pos) are the chunks of the K condition keys (prefix is a multiple of 16: launcher)
        p_pos = G == 1 ? 0 : __syncthreads_count(tid < chunk_total && chunk_id * 16 < min(a.prefix, n));
    } else {
        n_pos = (n + 15) >> 4;
        p_pos = G == 1 ? 0 : min((min(a.prefix, n) + 15) >> 4, n_pos);
    }
    af_append_attend_store<DT, G, SP, STG>(a, bias_s, qkv_s, red, SP ? list_s : nullptr, n_pos, p_pos, n, head, b0, res_s, ldres, st, js);
}

// ----------------------------------------------------------------------------------------------------------------- ln1 + qkv + attention
// grid (H, B / G), 1024 threads.  Dynamic LDS: bias row [Lpad] | xn [G][D] | qkv [G][192] | red [16][G+1][66] | stat [16][G] | chunk list [Lpad/16 + 2] (uint16)
template <int DT, int G, int WT, bool SP>   // KV-cache storage (0 fp32, 1 fp16), sequences per workgroup, decode-weight storage (0 fp32, 1 fp16), block-sparse layout
__global__ __launch_bounds__(1024) void ar_attn_fused_kernel(ArAttnFusedArgs a) {
    constexpr int NW = AF_WAVES;
    extern __shared__ float smem[];
    const int D = a.D;
    float* bias_s = smem;
    float* xn_s = bias_s + a.Lpad;
    float* qkv_s = xn_s + G * D;
    float* red = qkv_s + G * 192;
    float* stat = red + NW * (G + 1) * 66;
    uint16_t* list_s = reinterpret_cast<uint16_t*>(stat + NW * G);
    char* stage = reinterpret_cast<char*>(list_s + ((a.Lpad / 16 + 2 + 7) & ~7));   // staged K/V pieces: [16 waves][stage_cap] KiB, 16-byte aligned

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int head = blockIdx.x, grp = blockIdx.y;
    const int b0 = grp * G;
    const int n = a.d_n ? *a.d_n + a.n : a.n;   // context length incl. the new key
    const int row = n - 1;
    const float sl2 = a.scale * kLog2eF;        // scores live in the base-2 domain

    // K/V rows staged in LDS while ln1 and the projection run (ArAttnFusedArgs::stage_cap; one sequence per workgroup, dense walk): every wave requests the leading whole
    // pipeline steps of ITS OWN share of the key walk - U K pieces + U V pieces of 1 KiB per step - by LDS-DMA into its region behind the sink, and the walk reads them
    // from there (Attend::run_staged).  HBM has nothing else to do for the 13 us of the prologue; the 128 KB the CU's 160 KB of LDS leave free are 40 % of a
    // (sequence, head)'s fp16 K/V rows at the mean context of a decode.
    constexpr bool STG = G == 1;
    using TS = KvRow<DT>;
    constexpr int KPI = 64 / TS::LPK, PIECE = NW * KPI, SU = WalkU<DT, G>::value, ROWB = 64 * (DT ? 2 : 4);
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const char* st_w = stage + (wave_u * a.stage_cap) * 1024;   // this wave's region
    // staged steps of this wave.  Dense walk: whole pipeline steps of rows that are in the cache at kernel start.  Chunk-list walk (SP): its leading list positions
    // wave, wave + 16, ... whose chunks lie WHOLLY below the new key - lane j holds the chunk of the wave's j-th position (one extra row load with the kernel's first)
    int js = 0, cid_lane = 0;
    if (STG && a.stage_cap > 0) {
        if (SP) {
            const SparseVis& v0 = a.vis;
            const uint16_t* crow = v0.chunks + (long)head * v0.chunks_head_stride + (long)(row / v0.blk) * v0.chunks_ld;
            const int total = crow[0], pos = wave_u + NW * lane;
            cid_lane = crow[1 + min(pos, max(v0.chunks_ld - 2, 0))];
            const bool ok = pos < total && 16 * cid_lane + 16 <= n - 1 && lane < a.stage_cap / (2 * SU);
            js = __builtin_popcountll(__builtin_amdgcn_ballot_w64(ok));   // (ascending list: the staged positions are a prefix)
        } else {
            js = min(((n - 1) / PIECE) / SU, a.stage_cap / (2 * SU));
        }
    }
    auto stage_issue = [&]() {   // request the pieces in LDS order (step-major: K pieces of the step, then its V pieces)
        const long wrow = (((long)b0 * a.H + head) * a.Lmax + (SP ? 0 : wave_u * KPI)) * ROWB + lane * 16;
        for (int q_iss = 0; q_iss < js * 2 * SU; ++q_iss) {
            const int j = q_iss / (2 * SU), r = q_iss % (2 * SU);
            const long koff = SP ? (long)(16 * __builtin_amdgcn_readlane(cid_lane, j) + (r % SU) * KPI) * ROWB : (long)(j * SU + r % SU) * (PIECE * ROWB);
            const char* src = reinterpret_cast<const char*>(r >= SU ? a.vcache : a.kcache) + wrow + koff;
            glds16_hidden_nt(src, (unsigned)__builtin_amdgcn_readfirstlane((int)lds_addr_of(st_w + q_iss * 1024)));
        }
    };

    AF_TRACE(0);
    // ---- every load that does not depend on another load is requested up front, in the order the results are needed (vmcnt waits are in order):
    //      x rows and ln1 gamma / beta -> first q/k/v weight rows -> bias and visibility row of this step
    const int tcol = min(tid, D - 1);   // threads beyond D load a valid element and ignore it (no predicated loads, see rowsrc_at)
    float xv[G];
#pragma unroll
    for (int g = 0; g < G; ++g) xv[g] = rowsrc_at(a.x, b0 + g, tcol);
    const float lw = a.ln_w[tcol], lb = a.ln_b[tcol];

    // q/k/v projection: wave w owns rows j = 12 w .. 12 w + 11 of the head's 192 (q | k | v) x 64 rows, fetched RB rows at a time into two register
    // buffers.  The 16 workgroups of a head (one per sequence, same XCD) read the same rows through that XCD's L2: each starts at a different batch
    // so that at any moment they pull on different lines / channels instead of queueing on one.
    // fp16 weight storage: 8 elements per 16-byte load, twice the rows per batch for the same registers
    constexpr int EL = WT ? 8 : 4, NCH = WT ? 2 : 4;                 // elements per lane and load; loads per row (D <= 1024)
    // Rows per batch: small enough that NOTHING spills.  A spilled register is not free here: 1024 threads x 256 workgroups park 1 MB per register in scratch and
    // read it back, per launch - round 2's three-row batches (G = 1) spilled 1 / 4 registers in the fp32 / fp16-cache variants, i.e. up to 8 MB of extra traffic
    // next to 22-44 MB of K/V; with two rows per batch the same-box A/B gives 1.617 -> 1.560 (fp32 cache), 1.306 -> 1.215 (fp16 cache) and 1.144 -> 1.122
    // ms/step (fp16 cache + weights), and the FETCH_SIZE counter no longer shows MORE traffic at density 0.35 than at density 1.
    constexpr int RB = (G >= 4 ? 1 : 2) * (WT ? 2 : 1), NB = 12 / RB;
    constexpr bool PRE2 = G == 1;   // both register buffers requested when the x rows have arrived (in front of the staged K/V pieces), the loop refills behind each product
    typedef typename std::conditional<WT == 1, half8_t, f32x4>::type WV;
    const int rot = grp % NB;
    auto wrow = [&](int bi, int r) { return wave * 12 + ((bi + rot) % NB) * RB + r; };
    auto wbase = [&](int j) { return ((long)(j >> 6) * D + head * 64 + (j & 63)) * D; };
    WV wb[2][RB][NCH];
    auto load_batch = [&](int bi, WV (&w)[RB][NCH]) {
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const long o = wbase(wrow(bi, r));
#pragma unroll
            for (int c = 0; c < NCH; ++c) {   // columns >= D: a clamped (valid) address, ignored by dot_batch
                const long e = o + min(c * 64 * EL + lane * EL, D - EL);
                if (WT) w[r][c] = *reinterpret_cast<const WV*>(reinterpret_cast<const _Float16*>(a.wqkv_h) + e);
                else w[r][c] = *reinterpret_cast<const WV*>(a.wqkv + e);
            }
        }
    };
    // bias row of this step with the visibility mask folded in (shared by every sequence and head of the step): through registers, stored after ln1
    constexpr int BR = 3;
    // (the launcher points absent tables at valid memory: vis.has_* / has_bias say whether the values count)
    const SparseVis& vis = a.vis;
    const uint8_t* keep_row = vis.allowed + (long)head * vis.allowed_head_stride + (long)row * vis.ldallowed;
    const uint8_t* lay_row = vis.lay + (long)head * vis.lay_head_stride + (long)(row / vis.blk) * vis.nb;
    const float* bias_row = a.bias + (long)row * a.ldbias;
    float braw[BR];
    uint8_t kraw[BR];
#pragma unroll
    for (int j = 0; j < BR; ++j) {   // raw values, clamped addresses: nothing below touches them before the statistics are done (a use would wait for the loads)
        const int k = min(tid + 1024 * j, n - 1);
        braw[j] = bias_row[k];
        kraw[j] = keep_row[k];
    }
    // (the block-layout row and the key-chunk list are only fetched when the layout actually hides something - density < 1 - and only after the projection:
    // held across it they push the fp32 variants over the 128-register budget of a 16-wave workgroup)
    const uint16_t* chunk_row = vis.chunks + (long)head * vis.chunks_head_stride + (long)(row / vis.blk) * vis.chunks_ld;


    // ---- ln1 FOLDED into the projection.  LN(x) W^T + b = rstd (x o gamma) W^T - rstd mean (W gamma) + (W beta + b): the dot products run on x o gamma, which exists the moment
    //      the x rows are in, and the row statistics are computed in their shadow; mean / rstd and the two per-row constants cs = W gamma, ds = W beta + b (launch_ar_ln_fold,
    //      once per layer) only enter a 192-value fix-up after the row loop.  The direct form - statistics (two barrier rounds), normalised row to LDS, barrier, THEN the first
    //      product - spent 2.8 us between the rows' arrival and the first multiply-add, most of it with every wave blocked at the request instructions of its first weight
    //      batches (a CU's address pipe takes 64 B per clock: 256 KB = 1.9 us) before it could even start the statistics.
    //      (Same arithmetic up to fp32 rounding of the re-associated sum: tokens equal to the oracle's on every fixture, tests/test_models_gpu.py.)
    //      Same-box A/Bs, ms per decode step: fp16 cache + fp16 weights 1.067 -> 1.050 when it was introduced; with fp32 weights it lost then (1.208 -> 1.225: the prologue
    //      was bound by the weight rows + staged pieces in the CU's request queue) and wins at the end of the round (1.103-1.105 -> 1.092, 1.398-1.408 -> 1.395-1.397): one form.
    const int jb = wave * 12 + min(lane, 11);
    float* stat2 = red;                         // scratch inside `red` (free until the key walk): second-pass sums | raw x, gamma, beta of this head's 64 columns
    float* xh_s = red + NW * G;
    float* gh_s = xh_s + G * 64;
    float c_mine, d_mine;
    {
        float s[G];
#pragma unroll
        for (int g = 0; g < G; ++g) s[g] = wave_sum_dpp(tid < D ? xv[g] : 0.f);
        if (lane == 0) {
#pragma unroll
            for (int g = 0; g < G; ++g) stat[wave * G + g] = s[g];
        }
        if (tid < D) {
#pragma unroll
            for (int g = 0; g < G; ++g) xn_s[g * D + tid] = xv[g] * lw;   // x o gamma: what the row loop multiplies
        }
        if ((tid >> 6) == head && tid < D) {
#pragma unroll
            for (int g = 0; g < G; ++g) xh_s[g * 64 + lane] = xv[g];
            gh_s[lane] = lw; gh_s[64 + lane] = lb;
        }
        lds_barrier();   // (from here to the walk every barrier orders LDS only: K/V pieces and weight rows stay in flight across them)
        AF_TRACE(6);
        // the x rows have arrived: only now request the first weight batches (issued earlier, the 50 MB the 256 workgroups ask their L2s for at
        // once would queue in front of the later workgroups' x rows)
        load_batch(0, wb[0]);
        if (PRE2) load_batch(1, wb[1]);   // (G > 1: two batches across the G rows' statistics spill)
        // this wave's 12 pairs of row constants, one per lane (applied after the row loop).  A load INSIDE the row loop sits, in the in-order return stream, behind the next
        // batch's weight loads just issued: every row then waited for the whole next batch - vmcnt(0) in front of each qkv_s store - and the double buffering was void
        // (the projection took 8 us with fp32 and with fp16 weights alike)
        c_mine = a.ln_cs[(long)(jb >> 6) * D + head * 64 + (jb & 63)];
        d_mine = a.ln_ds[(long)(jb >> 6) * D + head * 64 + (jb & 63)];
        // second statistics pass: in the shadow of the weight loads, read by the fix-up behind the row loop's closing barrier
        float mean[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) t += stat[w * G + g];
            mean[g] = t / (float)D;
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const float d = tid < D ? xv[g] - mean[g] : 0.f;
            s[g] = wave_sum_dpp(d * d);
        }
        if (lane == 0) {
#pragma unroll
            for (int g = 0; g < G; ++g) stat2[wave * G + g] = s[g];
        }
        AF_TRACE(7);
#pragma unroll
        for (int j = 0; j < BR; ++j)
            if (tid + 1024 * j < n)
                bias_s[tid + 1024 * j] = (kraw[j] || !vis.has_allowed) ? (a.has_bias ? braw[j] * sl2 : 0.f) : kNegBig;
        for (int k = tid + 1024 * BR; k < n; k += 1024)   // sequences longer than 3072: the remainder the plain way
            bias_s[k] = (keep_row[k] || !vis.has_allowed) ? (a.has_bias ? bias_row[k] * sl2 : 0.f) : kNegBig;
    }

    AF_TRACE(1);
    // ---- q/k/v projection of this head (folded form: on x o gamma)
    {
        auto dot_batch = [&](int bi, const WV (&w)[RB][NCH]) {
            float accp[RB][G];
#pragma unroll
            for (int r = 0; r < RB; ++r) {
#pragma unroll
                for (int g = 0; g < G; ++g) accp[r][g] = 0.f;
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    const int col = c * 64 * EL + lane * EL;
                    if (col < D) {
#pragma unroll
                        for (int g = 0; g < G; ++g) {
#pragma unroll
                            for (int e4 = 0; e4 < EL; e4 += 4) {
                                const float4 x4 = *reinterpret_cast<const float4*>(xn_s + g * D + col + e4);
                                accp[r][g] = fmaf((float)w[r][c][e4 + 0], x4.x, accp[r][g]);
                                accp[r][g] = fmaf((float)w[r][c][e4 + 1], x4.y, accp[r][g]);
                                accp[r][g] = fmaf((float)w[r][c][e4 + 2], x4.z, accp[r][g]);
                                accp[r][g] = fmaf((float)w[r][c][e4 + 3], x4.w, accp[r][g]);
                            }
                        }
                    }
                }
            }
            // the RB x G wave sums of the batch, four (or two) per reduction pass; the lane a total lands in stores it (raw sum: fixed up below)
            const int sel = ((lane >> 5) & 1) * 2 + ((lane >> 4) & 1);
            if constexpr (RB == 4) {
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const float t = wave_sum4(accp[0][g], accp[1][g], accp[2][g], accp[3][g]);
                    if ((lane & 15) == 0) qkv_s[g * 192 + wrow(bi, sel)] = t;
                }
            } else if constexpr (RB == 2) {
#pragma unroll
                for (int g = 0; g < G; g += 2) {
                    if (g + 1 < G) {   // rows x two sequences
                        const float t = wave_sum4(accp[0][g], accp[1][g], accp[0][g + 1], accp[1][g + 1]);
                        if ((lane & 15) == 0) qkv_s[(g + (sel >> 1)) * 192 + wrow(bi, sel & 1)] = t;
                    } else {
                        const float t = wave_sum2(accp[0][g], accp[1][g]);
                        if ((lane & 31) == 0) qkv_s[g * 192 + wrow(bi, lane >> 5)] = t;
                    }
                }
            } else {
                static_assert(RB == 1 && G % 4 == 0, "one row per batch: four sequences per reduction pass");
#pragma unroll
                for (int g = 0; g < G; g += 4) {
                    const float t = wave_sum4(accp[0][g], accp[0][g + 1], accp[0][g + 2], accp[0][g + 3]);
                    if ((lane & 15) == 0) qkv_s[(g + sel) * 192 + wrow(bi, 0)] = t;
                }
            }
        };

#pragma unroll
        for (int bi = 0; bi < NB; bi += 2) {
            if (!PRE2 && bi + 1 < NB) load_batch(bi + 1, wb[1]);
            dot_batch(bi, wb[0]);
            if (bi + 2 < NB) load_batch(bi + 2, wb[0]);
            if (bi + 1 < NB) {
                dot_batch(bi + 1, wb[1]);
                if (PRE2 && bi + 3 < NB) load_batch(bi + 3, wb[1]);
            }
        }
        // The staged K/V pieces are requested HERE, behind the last product of the projection.  What was measured about the placement (MI355X, same-box A/Bs,
        // profiles/r04_ab_kv_stage.txt): the CU returns loads in issue order across its waves and a wave blocks at a VMEM instruction while the CU's request queue is
        // full, so (a) pieces requested before the x rows delay them from 1.6 to 6.3 us; (b) one 128 KB burst when the rows have arrived holds every wave at its request
        // instructions for ~5 us (HBM feeds one CU 25 GB/s): the walk 5 us shorter, the prologue 3 us longer; (c) pieces between the row batches: the batch requested
        // before them is waited for with vmcnt(0) - the compiler does not count hidden requests - and so waits for them.  Requested here nothing waits on them but the
        // walk, whose own first loads queue behind them: HBM streams without a gap from the end of the projection on.  ms/step, fp16 cache: fp16 weights 1.029-1.034
        // (between batches 1.031-1.036, all early 1.070, no staging 1.110); fp32 weights 1.197-1.204 (1.225 / 1.203 / 1.241); density 0.35: 0.953 (0.962 / 0.999 / 1.017).
        if (STG) stage_issue();
        lds_barrier();   // every wave's raw sums, second-pass statistics and bias row are in LDS
        // fix-up: the wave's own 12 rows (lanes 0..11) and, by the first 64 G threads, the residual ln1(x) of this head's columns into the (now free) row buffer
        if (lane < 12 || tid < 64 * G) {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                float t1 = 0.f, t2 = 0.f;
#pragma unroll
                for (int w = 0; w < NW; ++w) { t1 += stat[w * G + g]; t2 += stat2[w * G + g]; }
                const float mean = t1 / (float)D, rstd = rsqrtf(t2 / (float)D + a.eps);
                if (lane < 12) qkv_s[g * 192 + jb] = rstd * (qkv_s[g * 192 + jb] - mean * c_mine) + d_mine;
                if ((tid >> 6) == g) xn_s[g * D + head * 64 + lane] = (xh_s[g * 64 + lane] - mean) * rstd * gh_s[lane] + gh_s[64 + lane];
            }
        }
    }
    lds_barrier();
    AF_TRACE(2);
    af_layout_attend<DT, G, SP, STG>(a, vis, lay_row, chunk_row, bias_s, qkv_s, red, list_s, n, head, b0, xn_s + head * 64, D, st_w + lane * 16, js);
}

// (A head-cooperative form of this kernel - the 16 sequence-workgroups of a head exchanging rows and 12-column q | k | v slices through the XCD's L2, so that the head's
// weight rows cross the L2 -> CU path once instead of 16 times - was built in round 5, parity-green, and measured slower: two serialized exchanges + the 64 KB row tile
// cost 7.7 us against the ~4.4 us the projection proper takes here (the rest of "ln1 done -> qkv done" is the K/V piece issue).  git history + EXPERIMENTS.md round 5.)

// ----------------------------------------------------------------------------------------------------------------- decode attention proper
// decode_path = split: the row's q | k | v were projected for the whole batch by the LayerNorm + QKV kernel (skinny_fused_kernel<LN, RS>: the weight matrix is read from
// HBM once per layer instead of once per sequence through L2), so this kernel is the K/V stream and nothing else: bias / visibility row and chunk list -> LDS,
// k / v appended, the walk over the visible 16-key chunks, merge, + ln1(x) residual.  grid (H, B / G), 1024 threads.
// Dynamic LDS: bias row [Lpad] | residual [G][64] | qkv [G][192] | red [16][G+1][66] | chunk list [Lpad/16 + 2] (uint16)
template <int DT, int G, bool SP>
__global__ __launch_bounds__(1024) void ar_attn_kernel(ArAttnFusedArgs a) {
    constexpr int NW = AF_WAVES;
    extern __shared__ float smem[];
    const int D = a.D;
    float* bias_s = smem;
    float* res_s = bias_s + a.Lpad;
    float* qkv_s = res_s + G * 64;
    float* red = qkv_s + G * 192;
    uint16_t* list_s = reinterpret_cast<uint16_t*>(red + NW * (G + 1) * 66);

    const int tid = threadIdx.x;
    const int head = blockIdx.x, grp = blockIdx.y;
    const int b0 = grp * G;
    const int n = a.d_n ? *a.d_n + a.n : a.n;   // context length incl. the new key
    const int row = n - 1;
    const float sl2 = a.scale * kLog2eF;
    AF_TRACE(0);
    // all loads up front, unconditional on clamped addresses (see rowsrc_at)
    const int tq = min(tid, 192 * G - 1), gq = tq / 192, jq = tq % 192;
    const float qraw = a.qkv[(long)(b0 + gq) * 3 * D + (long)(jq >> 6) * D + head * 64 + (jq & 63)];
    const int tr = min(tid, 64 * G - 1);
    const float rraw = a.xn[(long)(b0 + (tr >> 6)) * D + head * 64 + (tr & 63)];
    constexpr int BR = 3;
    const SparseVis& vis = a.vis;
    const uint8_t* keep_row = vis.allowed + (long)head * vis.allowed_head_stride + (long)row * vis.ldallowed;
    const uint8_t* lay_row = vis.lay + (long)head * vis.lay_head_stride + (long)(row / vis.blk) * vis.nb;
    const float* bias_row = a.bias + (long)row * a.ldbias;
    float braw[BR];
    uint8_t kraw[BR], lraw[BR];
#pragma unroll
    for (int j = 0; j < BR; ++j) {
        const int k = min(tid + 1024 * j, n - 1);
        braw[j] = bias_row[k];
        kraw[j] = keep_row[k];
        lraw[j] = lay_row[k / vis.blk];
    }
    const uint16_t* chunk_row = vis.chunks + (long)head * vis.chunks_head_stride + (long)(row / vis.blk) * vis.chunks_ld;
    const int chunk_total = chunk_row[0];
    const int chunk_id = chunk_row[min(1 + tid, max(vis.chunks_ld - 1, 0))];

    if (tid < 192 * G) qkv_s[tid] = qraw;
    if (tid < 64 * G) res_s[tid] = rraw;
#pragma unroll
    for (int j = 0; j < BR; ++j)
        if (tid + 1024 * j < n)
            bias_s[tid + 1024 * j] = ((kraw[j] || !vis.has_allowed) && (lraw[j] || !vis.has_lay)) ? (a.has_bias ? braw[j] * sl2 : 0.f) : kNegBig;
    for (int k = tid + 1024 * BR; k < n; k += 1024)
        bias_s[k] = ((keep_row[k] || !vis.has_allowed) && (lay_row[k / vis.blk] || !vis.has_lay)) ? (a.has_bias ? bias_row[k] * sl2 : 0.f) : kNegBig;
    if (SP && tid < chunk_total) list_s[tid] = (uint16_t)chunk_id;
    __syncthreads();
    const int n_pos = SP ? __syncthreads_count(tid < chunk_total && chunk_id * 16 < n) : (n + 15) >> 4;
    const int p_pos = G == 1 ? 0 : (SP ? __syncthreads_count(tid < chunk_total && chunk_id * 16 < min(a.prefix, n)) : min((min(a.prefix, n) + 15) >> 4, n_pos));
    AF_TRACE(2);
    af_append_attend_store<DT, G, SP>(a, bias_s, qkv_s, red, SP ? list_s : nullptr, n_pos, p_pos, n, head, b0, res_s, 64);
}

#undef AF_TRACE

size_t ar_attn_lds_bytes(int G, int Lpad) {
    return ((size_t)Lpad + (size_t)G * 64 + (size_t)G * 192 + (size_t)AF_WAVES * (G + 1) * 66) * sizeof(float) + ((size_t)Lpad / 16 + 2 + 8) * sizeof(uint16_t) + 1024;
}

size_t ar_attn_fused_lds_bytes(int G, int D, int Lpad) {
    return ((size_t)Lpad + (size_t)G * D + (size_t)G * 192 + (size_t)AF_WAVES * (G + 1) * 66 + (size_t)AF_WAVES * G) * sizeof(float) + ((size_t)Lpad / 16 + 2 + 8) * sizeof(uint16_t) +
           1024;   // (+ 1 KiB of slack)
}

// LDS a workgroup may allocate on the CURRENT device (gfx950: 160 KB), queried once per device: one process may drive contexts on several GPUs, and both this limit
// and the > 64 KB opt-in below (hipFuncSetAttribute) are per-device state
constexpr int MAX_DEVICES = 64;
static int current_device_slot() {
    int dev = 0;
    (void)hipGetDevice(&dev);
    return dev >= 0 && dev < MAX_DEVICES ? dev : 0;
}
size_t ar_attn_fused_max_lds() {
    static std::atomic<size_t> v[MAX_DEVICES];
    const int dev = current_device_slot();
    size_t got = v[dev].load(std::memory_order_relaxed);
    if (!got) {
        int optin = 0;
        if (hipDeviceGetAttribute(&optin, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess || optin <= 0) optin = 64 * 1024;
        got = (size_t)optin;
        v[dev].store(got, std::memory_order_relaxed);
    }
    return got;
}

__global__ __launch_bounds__(256) void ar_ln_fold_kernel(const float* __restrict__ W, const float* __restrict__ b, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         float* __restrict__ cs, float* __restrict__ ds, int N, int K) {
    const int lane = threadIdx.x & 63, j = blockIdx.x * 4 + (threadIdx.x >> 6);   // one wave per row
    if (j >= N) return;
    double c = 0.0, d = 0.0;
    for (int k = lane; k < K; k += 64) {
        const double w = W[(long)j * K + k];
        c += w * (double)gamma[k];
        d += w * (double)beta[k];
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { c += __shfl_xor(c, o, 64); d += __shfl_xor(d, o, 64); }
    if (lane == 0) { cs[j] = (float)c; ds[j] = (float)(d + (double)b[j]); }
}
void launch_ar_ln_fold(const float* W, const float* b, const float* gamma, const float* beta, float* cs, float* ds, int N, int K, hipStream_t s) {
    hipLaunchKernelGGL(ar_ln_fold_kernel, dim3(cdiv(N, 4)), dim3(256), 0, s, W, b, gamma, beta, cs, ds, N, K);
    LAUNCH_CHECK();
}

bool ar_attn_fused_supported(int B, int G, int D, int H) { return D == H * 64 && D % 4 == 0 && D <= 1024 && (G == 1 || G == 2 || G == 4) && B % G == 0; }

void launch_ar_attn_fused(const ArAttnFusedArgs& a0, hipStream_t s) {
    ArAttnFusedArgs a = a0;
    BG_REQUIRE(ar_attn_fused_supported(a.B, a.G, a.D, a.H), "fused decode attention: unsupported shape B=%d G=%d D=%d H=%d", a.B, a.G, a.D, a.H);
    BG_REQUIRE(a.x.ns <= ROWSRC_MAX_SPLITS, "fused decode attention: at most %d partial sums per row", ROWSRC_MAX_SPLITS);
    a.Lpad = (int)round_up(a.Lmax, 4);
    const bool pre = a.qkv != nullptr;
    BG_REQUIRE(!pre || a.xn, "decode attention: precomputed q/k/v rows need the ln1(x) rows for the residual");
    BG_REQUIRE(pre || (a.ln_cs && a.ln_ds), "fused decode attention: the folded LayerNorm needs its row constants (launch_ar_ln_fold)");
    if (pre) { a.x = RowSrc{}; a.x.base = a.qkv; a.x.ld = 3 * a.D; }
    a.x = rowsrc_fix(a.x);
    a.has_bias = a.bias != nullptr;
    a.status = status_current();
    a.vis = vis_fix(a.vis, a.x.base);
    if (!a.bias) { a.bias = a.x.base; a.ldbias = 0; }
    BG_REQUIRE(a.G == 1 || a.prefix % 16 == 0, "fused decode attention: a shared prefix must be a multiple of 16 keys (prefix=%d)", a.prefix);
    BG_REQUIRE(!a.vis.has_chunks || a.vis.chunks_ld <= 1025, "fused decode attention: at most 1024 key chunks per row");
    size_t lds = pre ? ar_attn_lds_bytes(a.G, a.Lpad) : ar_attn_fused_lds_bytes(a.G, a.D, a.Lpad);
    BG_REQUIRE(lds <= 64 * 1024, "fused decode attention: %zu bytes of LDS needed (sequence length %d too long)", lds, a.Lmax);
    // K/V staging (fused kernel, G = 1, dense walk): what the CU's LDS has left beyond the kernel's own 24 KB, in whole pipeline steps per wave (gfx950: 160 KB per
    // workgroup -> 8 pieces per wave = 128 KB).  $BEVGEN_KV_STAGE overrides the launcher's choice (A/B switch; 0 = off)
    if (!pre && a.G == 1) {
        static const int env_cap = getenv("BEVGEN_KV_STAGE") ? atoi(getenv("BEVGEN_KV_STAGE")) : -1;
        const int step_pieces = 2 * (a.kv_dtype == 0 ? 4 : 2);   // K + V pieces of one pipeline step (WalkU)
        const int room = (int)((ar_attn_fused_max_lds() - 1024 - lds) / (AF_WAVES * 1024));   // (1 KiB kept back: the block-sparse variants hold 256 B of static LDS)
        int cap = a.stage_cap >= 0 ? a.stage_cap : (env_cap >= 0 ? env_cap : 8);
        cap = std::max(0, std::min(cap, room)) / step_pieces * step_pieces;
        a.stage_cap = cap;
        lds += (size_t)cap * AF_WAVES * 1024;
    } else {
        a.stage_cap = 0;
    }
    BG_REQUIRE(a.ksplit >= 1 && (a.ksplit == 1 || (pre && a.G == 1 && a.kws)), "decode attention: a key split needs the attention-only kernel, one sequence per workgroup and a workspace");
    dim3 grid(a.H, a.B / a.G, a.ksplit);
    // algorithmic bytes of one launch: K and V rows of the context, once each; the shared prefix once per group
    const double n_host = a.d_n ? a.n + a.n_hint : a.n;
    const double eb = a.kv_dtype == 0 ? 4 : 2;
    const double pfx = a.G > 1 ? (double)a.prefix : 0.0;
    // SP instantiations only when a layout hides something (density < 1): chunk lists are then always present (context.cpp / the operator entry build both)
    const bool sp = a.vis.has_chunks;
    BG_REQUIRE(sp || !a.vis.has_lay, "fused decode attention: a block layout needs its chunk lists");
    BG_REQUIRE(pre || !a.wqkv_h || a.D % 8 == 0, "fused decode attention: fp16 weights need D %% 8 == 0");
    // (every argument check is above: a ProfScope whose launch never happens would leave an unrecorded event pair behind)
    ProfScope prof(PROF_DECODE_ATTN, 2.0 * a.H * 64 * eb * ((double)a.B * (n_host - pfx) + (double)(a.B / a.G) * pfx), s, a.ksplit == 1);   // (one launch: events attached to it)
    const int dev_slot = current_device_slot();
    // (more than 64 KB of dynamic LDS has to be allowed per kernel function AND per device, once each; the flags are atomics: contexts may launch from several host threads)
#define AF_LAUNCH1(K) do { static std::atomic<bool> big[MAX_DEVICES]; if (lds > 64 * 1024 && !big[dev_slot].load(std::memory_order_acquire)) { HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&K), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ar_attn_fused_max_lds() - 1024)); big[dev_slot].store(true, std::memory_order_release); } \
                           if (prof.attached()) hipExtLaunchKernelGGL(K, grid, dim3(1024), lds, s, prof.ev_a(), prof.ev_b(), 0, a); \
                           else hipLaunchKernelGGL(K, grid, dim3(1024), lds, s, a); } while (0)
#define AF_LAUNCH(DT, GG, WW) do { if (sp) AF_LAUNCH1((ar_attn_fused_kernel<DT, GG, WW, true>)); else AF_LAUNCH1((ar_attn_fused_kernel<DT, GG, WW, false>)); } while (0)
#define AF_LAUNCH_G(DT, WW) do { if (a.G == 1) AF_LAUNCH(DT, 1, WW); else if (a.G == 2) AF_LAUNCH(DT, 2, WW); else AF_LAUNCH(DT, 4, WW); } while (0)
    if (pre) {
#define AP_LAUNCH1(K) do { if (prof.attached()) hipExtLaunchKernelGGL(K, grid, dim3(1024), lds, s, prof.ev_a(), prof.ev_b(), 0, a); \
                           else hipLaunchKernelGGL(K, grid, dim3(1024), lds, s, a); } while (0)
#define AP_LAUNCH(DT, GG) do { if (sp) AP_LAUNCH1((ar_attn_kernel<DT, GG, true>)); else AP_LAUNCH1((ar_attn_kernel<DT, GG, false>)); } while (0)
#define AP_LAUNCH_G(DT) do { if (a.G == 1) AP_LAUNCH(DT, 1); else if (a.G == 2) AP_LAUNCH(DT, 2); else AP_LAUNCH(DT, 4); } while (0)
        if (a.kv_dtype == 0) AP_LAUNCH_G(0); else AP_LAUNCH_G(1);
#undef AP_LAUNCH_G
#undef AP_LAUNCH
#undef AP_LAUNCH1
    } else if (a.wqkv_h) {
        if (a.kv_dtype == 0) AF_LAUNCH_G(0, 1); else AF_LAUNCH_G(1, 1);
    } else {
        if (a.kv_dtype == 0) AF_LAUNCH_G(0, 0); else AF_LAUNCH_G(1, 0);
    }
#undef AF_LAUNCH_G
#undef AF_LAUNCH
#undef AF_LAUNCH1
    LAUNCH_CHECK();
    if (a.ksplit > 1) {   // merge the key ranges: o / l + ln1(x)
        DecodeAttnArgs c;
        c.R = a.xn; c.ldr = a.D; c.O = a.out; c.ldo = a.ldo; c.B = a.B; c.H = a.H;
        launch_decode_attention_combine(c, a.kws, a.ksplit, s);
    }
}

// ----------------------------------------------------------------------------------------------------------------- (ln +) skinny GEMM
// C[M, N] = act(LN?(A)[M, K] W[N, K]^T + bias), M <= 64.  One workgroup = 16 output columns x one K slice of <= 1024; 8 waves split the slice.
// The weight slice (64 KB per workgroup) goes straight from HBM into registers with non-temporal 16-byte loads issued first; while it is in
// flight the workgroup builds its A tile in LDS: rows are fetched,
// optionally LayerNorm-ed (two-pass statistics, 32 values per lane, lanes -> waves through LDS), and stored k-chunk-major
// ([k/4][16 rows][4]) so that both the staging stores and the MFMA operand reads are contiguous 1 KiB per wave (conflict free).
// v_mfma_f32_16x16x4_f32 (exact fp32): the four k-slots of one MFMA are the four 16-lane quarters, quarter q owns k = 16c + 4q .. +3.
constexpr int SF_WAVES = 8;

__device__ __forceinline__ float4 ldg_nt4(const float* p) {
    const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
    return make_float4(v[0], v[1], v[2], v[3]);
}

// FD (with LN, plain A): the LayerNorm FOLDED into the product, as in the fused attention kernel: LN(x) W^T + b = rstd ((x o gamma) W^T - mean (W gamma)) + (W beta + b).
//     The raw rows go to LDS the moment they arrive and gamma is multiplied in where the MFMA operand is read; the two-pass row statistics no longer stand between the
//     rows' arrival and the first MFMA (two barrier rounds + the normalisation: 4.9 us from launch to "A tile staged" against 2.9 us in the MLP-down launch that has no
//     LayerNorm) - the second pass runs behind the A barrier and mean / rstd only enter the epilogue, with the per-column constants cs = W gamma, ds = W beta + b
//     (launch_ar_ln_fold, once per layer).
template <bool LN, int WT, bool RS = false, bool FD = false>   // WT: weight storage of the packed image, 0 fp32 ([K/16][64 lanes][4]), 1 fp16 ([K/32][64 lanes][8]); RS: A is a row source
__global__ __launch_bounds__(SF_WAVES * 64) void skinny_fused_kernel(SkinnyFusedArgs g) {
    static_assert(!FD || (LN && !RS), "the folded LayerNorm exists for the plain-A LayerNorm form");
    __shared__ float4 As[256 * 16];
    __shared__ float4 gm_s[FD ? 256 : 1];   // gamma (FD): read behind the A barrier, while other waves may already write `red`
    __shared__ float red[SF_WAVES][4][64];
    __shared__ float stat[2][SF_WAVES][16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int n0 = blockIdx.x * 16, split = blockIdx.y;
    const int kw = g.K / g.ksplit, kbase = split * kw, kper = kw / SF_WAVES;   // kper: multiple of 16, <= 128
    const int nch = kw >> 2;

#define SF_TRACE(i) do { if (g.trace && tid == 0) g.trace[(long)(blockIdx.y * gridDim.x + blockIdx.x) * 8 + (i)] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)
    SF_TRACE(0);
    // Loads are requested in the order their results are needed: the first A row chunk and the LayerNorm parameters, THEN the weight slice, so that
    // the statistics are computed while the 64 KB of weights are still in flight (vmcnt waits are in order per wave).
    auto load_a = [&](int mc, float4 (&v)[8]) {
        const int m = mc * 16 + r;
        const int mr = min(m, g.M - 1);
        if (RS) {
            // previous layer's split-K partials (index order), + bias, + residual: the order rowsrc_at uses.  Six loads per element: fetched in four passes of
            // two chunks (12 x 16 B in flight per thread) - all eight at once need 192 registers and spill next to the weight slice already in flight
            const RowSrc& rs = g.src;
            constexpr int PJ = 2;
            // wave-uniform bases + ONE 32-bit element offset per chunk (48 full 64-bit lane addresses kept live across the row-chunk loop were the spill)
            const float* pb[ROWSRC_RS_SPLITS];
#pragma unroll
            for (int k = 0; k < ROWSRC_RS_SPLITS; ++k) pb[k] = rs.partial + (long)(k < rs.ns ? k : 0) * rs.pstride;
#pragma unroll
            for (int half = 0; half < 8 / PJ; ++half) {
                float4 p[PJ][ROWSRC_RS_SPLITS], b[PJ], x[PJ];
#pragma unroll
                for (int jj = 0; jj < PJ; ++jj) {
                    const int c = min(q + 4 * wave + 32 * (half * PJ + jj), nch - 1);   // clamped, never predicated (see rowsrc_at)
                    const unsigned col = (unsigned)(kbase + 4 * c);
                    const unsigned po = (unsigned)mr * (unsigned)rs.pld + col, xo = (unsigned)mr * (unsigned)rs.ld + col;
#pragma unroll
                    for (int k = 0; k < ROWSRC_RS_SPLITS; ++k) p[jj][k] = *reinterpret_cast<const float4*>(pb[k] + po);
                    b[jj] = *reinterpret_cast<const float4*>(rs.bias + (rs.has_bias ? col : 0u));
                    x[jj] = *reinterpret_cast<const float4*>(rs.base + xo);
                }
#pragma unroll
                for (int jj = 0; jj < PJ; ++jj) {
                    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int k = 0; k < ROWSRC_RS_SPLITS; ++k) {   // absent partials alias valid memory (rowsrc_fix): loaded, not counted
                        sum.x += k < rs.ns ? p[jj][k].x : 0.f; sum.y += k < rs.ns ? p[jj][k].y : 0.f; sum.z += k < rs.ns ? p[jj][k].z : 0.f; sum.w += k < rs.ns ? p[jj][k].w : 0.f;
                    }
                    v[half * PJ + jj] = make_float4((sum.x + (rs.has_bias ? b[jj].x : 0.f)) + x[jj].x, (sum.y + (rs.has_bias ? b[jj].y : 0.f)) + x[jj].y,
                                                   (sum.z + (rs.has_bias ? b[jj].z : 0.f)) + x[jj].z, (sum.w + (rs.has_bias ? b[jj].w : 0.f)) + x[jj].w);
                }
                asm volatile("" ::: "memory");   // keep the next pass's loads behind this pass's sums (the scheduler would otherwise issue all 48 at once)
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = min(q + 4 * wave + 32 * j, nch - 1);   // clamped, never predicated (see rowsrc_at); surplus chunks / rows are ignored below
                v[j] = *reinterpret_cast<const float4*>(g.A + (long)mr * g.lda + kbase + 4 * c);
            }
        }
    };
    float4 v[8];
    load_a(0, v);
    // LayerNorm gamma / beta (K <= 1024 floats each): ONE float4 per thread, parked in LDS behind the first statistics barrier and read back at the normalisation.
    // (Eight float4 of each per thread were 128 KB of L2 loads per workgroup in front of the weight requests - loads go out and come back in order - 1.4 us of the CU's
    // address pipe: the weight slice arrived at 5.7 us in this kernel against 3.7-4.0 us in the MLP-down launch that has no LayerNorm.)
    float4 gb_raw = make_float4(0.f, 0.f, 0.f, 0.f);
    if (LN) {
        const int c = min(tid & 255, nch - 1);
        gb_raw = *reinterpret_cast<const float4*>((tid < 256 ? g.ln_w : g.ln_b) + 4 * c);   // launcher: a missing beta aliases gamma, has_ln_b = 0
    }
    float4 (*gb_s)[256] = reinterpret_cast<float4 (*)[256]>(&red[0][0][0]);   // gamma | beta share the 8 KB of `red` (free until the MFMAs are done)
    // epilogue constants of this thread's output column, requested here: fetched in the epilogue they put one more memory round trip behind the last MFMA
    float e_bias = 0.f, e_cs = 0.f, e_ds = 0.f;
    if (tid < 256) {
        const int col = min(n0 + (tid & 15), g.N - 1);
        if (FD) { e_cs = g.ln_cs[col]; e_ds = g.ln_ds[col]; }
        else if (g.ksplit == 1 && g.bias) e_bias = g.bias[col];
    }
    float4 wv[WT ? 1 : 8];
    half8_t wh[WT ? 4 : 1];
    if (WT) {
        // fp16 image: one wave load = 1 KiB = a 32-wide k-chunk of the 16 columns; lane (r, q) holds W[16 tile + r][32 chunk + 8 q .. + 7]
        const _Float16* wp = reinterpret_cast<const _Float16*>(g.Wp) + (((long)blockIdx.x * (g.K >> 5) + ((kbase + wave * kper) >> 5)) * 64 + lane) * 8;
#pragma unroll
        for (int u = 0; u < 4; ++u) wh[u] = __builtin_nontemporal_load(reinterpret_cast<const half8_t*>(wp + (long)min(u, (kper >> 5) - 1) * 512));
    } else {
        // packed layout (launch_pack_skinny_weight): one wave load = 1 KiB contiguous, the 8 loads of a wave 8 KiB, the workgroup's slice 64 KiB
        const float* wp = g.Wp + (((long)blockIdx.x * (g.K >> 4) + ((kbase + wave * kper) >> 4)) * 64 + lane) * 4;
#pragma unroll
        for (int u = 0; u < 8; ++u) wv[u] = ldg_nt4(wp + (long)min(u, (kper >> 4) - 1) * 256);   // u >= kper / 16: a repeat, not used
    }

    // (row-source form: one row chunk per launch - the launcher loops - so that no fold address has to live across a loop)
    const int n_mc = RS ? 1 : (g.M + 15) / 16;
    for (int mc = 0; mc < n_mc; ++mc) {
        if (mc > 0) load_a(mc, v);
        if (LN && FD) {
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (q + 4 * wave + 32 * j < nch) s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
            s += xor16(s);
            s += xor32(s);
            if (q == 0) stat[0][wave][r] = s;
            if (tid < 256) gm_s[tid] = gb_raw;
        } else if (LN) {
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (q + 4 * wave + 32 * j < nch) s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
            s += xor16(s);
            s += xor32(s);
            if (q == 0) stat[0][wave][r] = s;
            gb_s[tid >> 8][tid & 255] = gb_raw;   // (every row chunk: `red` is rewritten by the chunk's reduction)
            // LDS-only barriers up to the MFMAs: __syncthreads() also waits for every load in flight, i.e. it parks the statistics behind the weight slice
            // (requested above, 64 KB per workgroup from HBM) instead of running them in its shadow
            lds_barrier();
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < SF_WAVES; ++w) t += stat[0][w][r];
            const float mean = t / (float)g.K;
            float qq = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (q + 4 * wave + 32 * j < nch) {
                    const float a0 = v[j].x - mean, a1 = v[j].y - mean, a2 = v[j].z - mean, a3 = v[j].w - mean;
                    qq += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
                }
            }
            qq += xor16(qq);
            qq += xor32(qq);
            if (q == 0) stat[1][wave][r] = qq;
            lds_barrier();
            t = 0.f;
#pragma unroll
            for (int w = 0; w < SF_WAVES; ++w) t += stat[1][w][r];
            const float rstd = rsqrtf(t / (float)g.K + g.eps);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float fb = g.has_ln_b ? 1.f : 0.f;
                const int c = min(q + 4 * wave + 32 * j, nch - 1);
                const float4 gm = gb_s[0][c], bt = gb_s[1][c];
                v[j] = make_float4(fmaf(bt.x, fb, (v[j].x - mean) * rstd * gm.x), fmaf(bt.y, fb, (v[j].y - mean) * rstd * gm.y),
                                   fmaf(bt.z, fb, (v[j].z - mean) * rstd * gm.z), fmaf(bt.w, fb, (v[j].w - mean) * rstd * gm.w));
            }
            // every workgroup holds the same normalised rows: workgroup i writes the chunks c with c % gridDim.x == i (each element once)
            if (RS && g.xn_out) {   // (only the row-source form - the LayerNorm + QKV projection of the split decode path - keeps ln1(x))
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int c = q + 4 * wave + 32 * j, m = mc * 16 + r;
                    if (c < nch && m < g.M && c % (int)gridDim.x == (int)blockIdx.x) *reinterpret_cast<float4*>(g.xn_out + (long)m * g.ldxn + 4 * c) = v[j];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = q + 4 * wave + 32 * j;
            if (c < nch) As[c * 16 + r] = v[j];
        }
        lds_barrier();
        SF_TRACE(1);
        // FD: everything that does not need the weight slice runs in front of the wait for it - the second pass of the statistics from the rows still in registers
        // (consumed by the epilogue, behind the `red` barrier) and the MFMA A operands x o gamma (k-chunk c4 of the row; LayerNorm launches have one K slice).
        // (gamma through LDS, one float4 per thread: fetched per operand chunk into registers - 8 more loads per thread, behind the weight requests - the same-box A/B
        // gave 0.998-1.001 against 0.989-0.994 ms/step: the weight slice is the long pole of this launch and everything else in the CU's request queue delays it)
        float4 aop[FD ? 8 : 1];
        if (FD) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < SF_WAVES; ++w) t += stat[0][w][r];
            const float mean = t / (float)g.K;
            float qq = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (q + 4 * wave + 32 * j < nch) {
                    const float a0 = v[j].x - mean, a1 = v[j].y - mean, a2 = v[j].z - mean, a3 = v[j].w - mean;
                    qq += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
                }
            }
            qq += xor16(qq);
            qq += xor32(qq);
            if (q == 0) stat[1][wave][r] = qq;
#pragma unroll
            for (int i = 0; i < 8; ++i) {   // operand i of the MFMA loop below (chunks past the wave's slice: clamped, not used)
                const int c4 = min(WT ? ((wave * kper + 32 * (i >> 1)) >> 2) + 2 * q + (i & 1) : ((wave * kper + 16 * i) >> 2) + q, nch - 1);
                aop[i] = mul4(As[c4 * 16 + r], gm_s[c4]);
            }
        }
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (WT) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (32 * u < kper) {   // quarter q owns k = 32 c + 8 q .. + 7 of the chunk: two A float4 (chunks of 4 k), eight MFMAs
                    const int c4 = ((wave * kper + 32 * u) >> 2) + 2 * q;
                    const float4 a0 = FD ? aop[FD ? 2 * u : 0] : As[c4 * 16 + r], a1 = FD ? aop[FD ? 2 * u + 1 : 0] : As[(c4 + 1) * 16 + r];
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, (float)wh[u][0], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, (float)wh[u][1], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, (float)wh[u][2], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, (float)wh[u][3], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, (float)wh[u][4], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, (float)wh[u][5], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, (float)wh[u][6], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, (float)wh[u][7], acc, 0, 0, 0);
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (16 * u < kper) {
                    const float4 a4 = FD ? aop[FD ? u : 0] : As[(((wave * kper + 16 * u) >> 2) + q) * 16 + r];
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, wv[u].x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, wv[u].y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, wv[u].z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, wv[u].w, acc, 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) red[wave][j][lane] = acc[j];
        __syncthreads();
        SF_TRACE(2);
        if (tid < 256) {   // one output element per thread: the 8 waves' partial sums in wave order, then the epilogue
            // C/D layout of the 16x16 MFMA: col = lane & 15, row = 4 * (lane >> 4) + reg; thread t finishes (reg j = t >> 6, lane t & 63)
            const int j = tid >> 6, ln = tid & 63;
            float o = 0.f;
#pragma unroll
            for (int w = 0; w < SF_WAVES; ++w) o += red[w][j][ln];
            const int col = n0 + (ln & 15), mo = mc * 16 + 4 * (ln >> 4) + j;
            if (col < g.N && mo < g.M) {
                if (g.ksplit > 1) {
                    g.C[((long)split * g.M + mo) * g.N + col] = o;
                } else {
                    if (FD) {
                        const int rw = 4 * (ln >> 4) + j;
                        float t1 = 0.f, t2 = 0.f;
#pragma unroll
                        for (int w = 0; w < SF_WAVES; ++w) { t1 += stat[0][w][rw]; t2 += stat[1][w][rw]; }
                        const float mean = t1 / (float)g.K, rstd = rsqrtf(t2 / (float)g.K + g.eps);
                        o = rstd * (o - mean * e_cs) + e_ds;
                    } else {
                        o += e_bias;
                    }
                    if (g.act == ACT_GELU) o = gelu_erf(o);
                    g.C[(long)mo * g.ldc + col] = o;
                }
            }
        }
        if (mc + 1 < n_mc) __syncthreads();   // As / red are rewritten by the next row chunk
    }
    SF_TRACE(3);
#undef SF_TRACE
}

// ----------------------------------------------------------------------------------------------------------------- both MLP projections in one launch
// ar_mlp_fused_kernel: ln2 (folded) + MLP up-projection + GELU, an exchange INSIDE one XCD, then the MLP down-projection - one launch instead of two per layer.
//
// Why an exchange is affordable here although round 3 measured 7.9 us for one (tools/gridbar/xchg_probe.hip): that one crossed XCDs - flag and data had to reach the memory
// side.  Workgroup i of a launch runs on XCD i % 8 (tools/xcc_probe; checked at run time below), so with the hidden column tiles dealt out as tile = 32 (i % 8) + i / 8 the
// 32 workgroups of an XCD produce one contiguous 512-column slice of the hidden row = exactly one K slice of the down-projection, and both sides of the exchange share that
// XCD's L2: producers use plain stores (the vector L1 is write-through: an acknowledged store is in L2), consumers plain loads of lines their CU has not read in this launch,
// and the rendezvous is one L2 counter per XCD.  Measured (tools/gridbar/xcd_xchg_probe.hip, profiles/r05_xcd_xchg_probe.txt): 1.1 us per such barrier, 2.3 us to read a
// 64 KB slice back, no agent-scope fence anywhere.  What the fusion buys: the down-projection's launch ramp and its cold weight burst (requested here at kernel start, 32 KB per
// workgroup, in registers long before the hidden slice exists) disappear behind the up-projection.
//
// Co-residency: every workgroup waits for its 31 XCD peers, so all of them must be resident - true whenever the launch has the GPU to itself (4 D / 16 = 256 workgroups of 8
// waves, one per CU; the decode step is a single chain of kernels on one stream).  Two such launches interleaved on one device (two contexts decoding at once) could starve each
// other: the spin is bounded (MLPF_TIMEOUT_TICKS of the 100 MHz clock) and a timeout raises the context's error word instead of hanging; the host serialises the decode
// steps of different contexts of one process on a device (ar.cpp) so that this does not happen in the first place.
//
// The barrier word is a monotonic arrival counter (one per XCD at sync[64 x], a 256-byte slot each): nothing has to be reset between launches, so the launch replays inside
// a hipGraph; sync[512 + i] records the XCD workgroup i saw.
// After a timeout the launch is POISONED (sync[MLPF_POISON] = 1): the decode steps already enqueued behind it (up to 2100 x 24 launches of a replayed graph) return at once
// instead of spinning 200 ms each; their output is garbage, which the status word has already said.  The host reads the word at its next synchronisation point
// (bevgen_synchronize / the next call), reports the call as failed and takes this context to the two-launch form (Ctx::check_status).
constexpr int MLPF_POISON = 1024;
size_t mlp_fused_sync_words() { return 512 + 1024; }

// fp16 weight storage: the product of 8 consecutive k of a row with the weight fragment on the f16 matrix pipe instead of eight v_mfma_f32_16x16x4_f32 (32 cycles of
// the SIMD's matrix pipe each, 40 as a dependent chain: the 32 + 32 of a workgroup's two projections were ~2 us of the launch's 10.7).  The fp32 activation is split
// a = hi + lo 2^-11 (two f16 numbers, as everywhere in Route M; the fp16 weight is exact), so a w = hi w + 2^-11 lo w with both products exact in the fp32 accumulators:
// two v_mfma_f32_16x16x32_f16 (~17 cycles each) on separate accumulators, merged by mlpf_merge.  Lane (r, q) holds row r, k = 8 q .. 8 q + 7 of both operands - the
// same (lane, element) -> k map on either side, which is all a matrix instruction needs.  Range: hi overflows for |a| >= 65520 -> inf / NaN, flagged (BG_ST_F16_RANGE).
// The phase-1 operands are the RAW residual-stream rows times gamma (the LayerNorm is folded in after the product: mean / rstd enter in the epilogue), so what has to stay
// below 65520 is |x gamma| - not the normalised value; phase 2's are the GELU outputs.
__device__ __forceinline__ void mlpf_mma_f16(const float4& a0, const float4& a1, const half8_t& w, f32x4& acc_hi, f32x4& acc_lo, unsigned& bad) {
    half8_t ah, al;
    const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) { ah[e] = split_hi(av[e]); al[e] = split_lo(av[e], ah[e]); }
    guard_half8(ah, bad);
    acc_hi = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, w, acc_hi, 0, 0, 0);
    acc_lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, w, acc_lo, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mlpf_merge(const f32x4& hi, const f32x4& lo) {
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = hi[j] + lo[j] * (1.f / 2048.f);
    return o;
}

template <int WT>   // weight storage of both packed images: 0 fp32, 1 fp16
__global__ __launch_bounds__(SF_WAVES * 64) void ar_mlp_fused_kernel(MlpFusedArgs g) {
    __shared__ float4 As[256 * 16];          // phase 1: the rows [k/4][16 rows] (K = D <= 1024); phase 2: the hidden slice (D / 2 columns)
    __shared__ float4 gm_s[256];             // ln2 gamma
    __shared__ float red[2][SF_WAVES][4][64];
    __shared__ float stat[2][SF_WAVES][16];
    __shared__ unsigned flag_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int D = g.D, K2 = 4 * D;
    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;     // workgroup jx of the per_xcd (= D / 32) on XCD xcd
    const int tile = xcd * per_xcd + jx;                                                 // hidden column tile: the XCD's tiles are contiguous
    const int n0 = tile * 16;
    const int nch = D >> 2, kper = D / SF_WAVES;                                         // phase 1: float4 chunks per row, K slice of a wave (128 at D = 1024)
    const int slice = K2 >> 3, kper2 = slice / SF_WAVES;                                 // phase 2: this XCD's K slice of the down-projection (512), per wave (64)
#define MF_TRACE(i) do { if (g.trace && tid == 0) g.trace[(long)blockIdx.x * 8 + (i)] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)
    MF_TRACE(0);
    unsigned bad = 0;   // an activation outside the f16 operand range (WT = 1)
    unsigned* cnt = g.sync + 64 * xcd;   // monotonic arrival counter of this XCD's workgroups (never reset: per_xcd arrivals per launch, compared modulo 2^32)
    // ---- requests, in the order the results are needed: rows, gamma, row constants, up weights, down weights
    const int n_mc = (g.M + 15) >> 4;   // row chunks of 16 (M <= 64): the weight slices stay in registers across them
    float4 v[8];
    auto load_rows = [&](int mc) {
        const int mr = min(16 * mc + r, g.M - 1);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = min(q + 4 * wave + 32 * j, nch - 1);   // clamped, never predicated (see rowsrc_at)
            v[j] = *reinterpret_cast<const float4*>(g.A + (long)mr * g.lda + 4 * c);
        }
    };
    load_rows(0);
    float4 gb_raw = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid < 256) gb_raw = *reinterpret_cast<const float4*>(g.ln_w + 4 * min(tid, nch - 1));
    float e_cs = 0.f, e_ds = 0.f;
    if (tid < 256) { e_cs = g.ln_cs[n0 + (tid & 15)]; e_ds = g.ln_ds[n0 + (tid & 15)]; }
    float4 wv[WT ? 1 : 8];
    half8_t wh[WT ? 4 : 1];
    float4 dv[WT ? 1 : 8];   // down weights: 2 column tiles x this wave's kper2 = 64 of the slice
    half8_t dh[WT ? 4 : 1];
    const int t2 = 2 * jx;   // this workgroup's two output column tiles of the down-projection
    if (WT) {
        const _Float16* wp = reinterpret_cast<const _Float16*>(g.Wup) + (((long)tile * (D >> 5) + ((wave * kper) >> 5)) * 64 + lane) * 8;
#pragma unroll
        for (int u = 0; u < 4; ++u) wh[u] = __builtin_nontemporal_load(reinterpret_cast<const half8_t*>(wp + (long)min(u, (kper >> 5) - 1) * 512));
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const _Float16* dp = reinterpret_cast<const _Float16*>(g.Wdn) + (((long)(t2 + t) * (K2 >> 5) + ((xcd * slice + wave * kper2) >> 5) + u) * 64 + lane) * 8;
                dh[2 * t + u] = __builtin_nontemporal_load(reinterpret_cast<const half8_t*>(dp));
            }
    } else {
        const float* wp = g.Wup + (((long)tile * (D >> 4) + ((wave * kper) >> 4)) * 64 + lane) * 4;
#pragma unroll
        for (int u = 0; u < 8; ++u) wv[u] = ldg_nt4(wp + (long)min(u, (kper >> 4) - 1) * 256);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float* dp = g.Wdn + (((long)(t2 + t) * (K2 >> 4) + ((xcd * slice + wave * kper2) >> 4) + u) * 64 + lane) * 4;
                dv[4 * t + u] = ldg_nt4(dp);
            }
    }
    // ---- phase 1: ln2 folded into the up-projection (the FD form of skinny_fused_kernel: raw rows to LDS, gamma at the operand read, statistics in the shadow of the weights)
    for (int mc = 0; mc < n_mc; ++mc) {
    if (mc > 0) load_rows(mc);
    {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (q + 4 * wave + 32 * j < nch) s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
        s += xor16(s);
        s += xor32(s);
        if (q == 0) stat[0][wave][r] = s;
        if (tid < 256 && mc == 0) gm_s[tid] = gb_raw;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = q + 4 * wave + 32 * j;
        if (c < nch) As[c * 16 + r] = v[j];
    }
    lds_barrier();
    MF_TRACE(1);
    float4 aop[8];
    {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < SF_WAVES; ++w) t += stat[0][w][r];
        const float mean = t / (float)D;
        float qq = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (q + 4 * wave + 32 * j < nch) {
                const float a0 = v[j].x - mean, a1 = v[j].y - mean, a2 = v[j].z - mean, a3 = v[j].w - mean;
                qq += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
            }
        }
        qq += xor16(qq);
        qq += xor32(qq);
        if (q == 0) stat[1][wave][r] = qq;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c4 = min(WT ? ((wave * kper + 32 * (i >> 1)) >> 2) + 2 * q + (i & 1) : ((wave * kper + 16 * i) >> 2) + q, nch - 1);
            aop[i] = mul4(As[c4 * 16 + r], gm_s[c4]);
        }
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (WT) {
        f32x4 acc_lo = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (32 * u < kper) mlpf_mma_f16(aop[2 * u], aop[2 * u + 1], wh[u], acc, acc_lo, bad);
        acc = mlpf_merge(acc, acc_lo);
    } else {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (16 * u < kper) {
                const float4 a4 = aop[u];
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, wv[u].x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, wv[u].y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, wv[u].z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, wv[u].w, acc, 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) red[0][wave][j][lane] = acc[j];
    lds_barrier();
    MF_TRACE(2);
    if (tid < 256) {   // one hidden element per thread: the 8 waves' partial sums in wave order, LayerNorm fix-up, GELU
        const int j = tid >> 6, ln = tid & 63;
        float o = 0.f;
#pragma unroll
        for (int w = 0; w < SF_WAVES; ++w) o += red[0][w][j][ln];
        const int rw = 4 * (ln >> 4) + j;
        float t1 = 0.f, t2s = 0.f;
#pragma unroll
        for (int w = 0; w < SF_WAVES; ++w) { t1 += stat[0][w][rw]; t2s += stat[1][w][rw]; }
        const float mean = t1 / (float)D, rstd = rsqrtf(t2s / (float)D + g.eps);
        o = gelu_erf(rstd * (o - mean * e_cs) + e_ds);
        if (16 * mc + rw < g.M) g.hidden[(long)(16 * mc + rw) * K2 + n0 + (ln & 15)] = o;
    }
    if (mc + 1 < n_mc) __syncthreads();   // As / red / stat are rewritten by the next row chunk
    }
    // ---- the exchange: this XCD's per_xcd workgroups have all stored their 16 hidden columns
    unsigned my_xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(my_xcc));
    my_xcc &= 0xf;
    if (tid == 0) g.sync[512 + blockIdx.x] = my_xcc;
    __builtin_amdgcn_s_waitcnt(0);   // (every wave: its stores are acknowledged, i.e. in L2)
    __syncthreads();
    if (tid == 0) {
        unsigned ok = 1;
        // One atomic per arrival and nothing to clean up: the counter only grows, launch k of this buffer's life takes it from k per_xcd to (k + 1) per_xcd (launches on
        // one buffer are serialised), so the value my own arrival returns tells me which multiple of per_xcd completes MY launch.  The last arrival IS the release - no
        // second word, no second round trip on the critical path - and the comparison is modulo 2^32 (134 million launches per wrap, and a wrap is harmless).
        const unsigned old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = (old / (unsigned)per_xcd + 1u) * (unsigned)per_xcd;
        if (old + 1u != target) {
            // (an earlier launch on this buffer timed out: do not spin again - the results are garbage either way, and the status word has said so.  Read only by a workgroup
            // that has to wait anyway: nothing on the fast path)
            const bool poisoned = __hip_atomic_load(g.sync + MLPF_POISON, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
            const long long t_in = __builtin_amdgcn_s_memrealtime();
            if (poisoned) ok = 0;
            else while ((int)(__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {   // (an L2-coherent read: the vector L1 never serves it)
                __builtin_amdgcn_s_sleep(1);
                if (__builtin_amdgcn_s_memrealtime() - t_in > MLPF_TIMEOUT_TICKS) { ok = 0; break; }
            }
        }
        flag_s = ok;
    }
    __syncthreads();
    MF_TRACE(3);
    if (tid < per_xcd && g.sync[512 + 8 * tid + xcd] != my_xcc) status_raise(g.err, BG_ST_MLP_PLACEMENT);   // a peer that is NOT on this XCD: its stores are not in this L2 (placement assumption broken)
    if (!flag_s) {
        if (tid == 0) {
            status_raise(g.err, BG_ST_MLP_BARRIER);
            __hip_atomic_store(g.sync + MLPF_POISON, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // acquire side of the exchange: the peers' hidden columns sit in this XCD's L2 (acknowledged stores, counted arrivals).  This CU's vector L1 could in principle still
    // hold lines of g.hidden from THIS workgroup's own phase-1 stores - it wrote 64 bytes of each 128-byte line it reads back below, the neighbouring tile owns the other
    // half.  g.acq: 1 = the slice is read with sc1 loads (served by the L2, whatever the L1 holds; the default), 2 = buffer_inv sc1 in front of plain loads (the
    // textbook agent-scope acquire: on this part it also drops the XCD's non-local L2 lines - measured +15 us per launch), 0 = plain loads (round 5)
    if (g.acq == 2) asm volatile("buffer_inv sc1" ::: "memory");
    // ---- phase 2: down-projection of this XCD's hidden slice, 2 x 16 output columns per workgroup, into partial plane `xcd`
    const int nch2 = slice >> 2;   // 128 chunks of 4
    for (int mc = 0; mc < n_mc; ++mc) {
    {
        const int mr = min(16 * mc + r, g.M - 1);
        f32x4 hv[4];
        if (g.acq == 1) {   // four sc1 loads in flight, ONE wait (the "+v" operands tie every later use of the values to the wait)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float* hp = g.hidden + (long)mr * K2 + xcd * slice + 4 * min(q + 4 * wave + 32 * j, nch2 - 1);
                asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(hv[j]) : "v"(hp) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(hv[0]), "+v"(hv[1]), "+v"(hv[2]), "+v"(hv[3]) : : "memory");
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) hv[j] = *reinterpret_cast<const f32x4*>(g.hidden + (long)mr * K2 + xcd * slice + 4 * min(q + 4 * wave + 32 * j, nch2 - 1));
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = q + 4 * wave + 32 * j;
            if (c < nch2) As[c * 16 + r] = make_float4(hv[j][0], hv[j][1], hv[j][2], hv[j][3]);
        }
    }
    __syncthreads();
    f32x4 acc2[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    if (WT) {
        f32x4 acc2_lo[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int c4 = ((wave * kper2 + 32 * u) >> 2) + 2 * q;
            const float4 a0 = As[c4 * 16 + r], a1 = As[(c4 + 1) * 16 + r];
            half8_t ah, al;   // (the split of the hidden values is shared by the two column tiles)
            const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) { ah[e] = split_hi(av[e]); al[e] = split_lo(av[e], ah[e]); }
            guard_half8(ah, bad);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                acc2[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, dh[2 * t + u], acc2[t], 0, 0, 0);
                acc2_lo[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, dh[2 * t + u], acc2_lo[t], 0, 0, 0);
            }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) acc2[t] = mlpf_merge(acc2[t], acc2_lo[t]);
    } else {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float4 a4 = As[(((wave * kper2 + 16 * u) >> 2) + q) * 16 + r];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const float4 w4 = dv[4 * t + u];
                acc2[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, w4.x, acc2[t], 0, 0, 0);
                acc2[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, w4.y, acc2[t], 0, 0, 0);
                acc2[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, w4.z, acc2[t], 0, 0, 0);
                acc2[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, w4.w, acc2[t], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) red[t][wave][j][lane] = acc2[t][j];
    __syncthreads();
    MF_TRACE(4);
    {   // thread (t, j, lane): one element of one of the two 16 x 16 output tiles
        const int t = tid >> 8, j = (tid >> 6) & 3, ln = tid & 63;
        float o = 0.f;
#pragma unroll
        for (int w = 0; w < SF_WAVES; ++w) o += red[t][w][j][ln];
        const int mo = 16 * mc + 4 * (ln >> 4) + j, col = (t2 + t) * 16 + (ln & 15);
        if (mo < g.M) g.C[((long)xcd * g.M + mo) * D + col] = o;
    }
    if (mc + 1 < n_mc) __syncthreads();   // As / red are rewritten by the next row chunk
    }
    if (WT && bad) status_raise(g.err, BG_ST_F16_RANGE);
    MF_TRACE(5);
#undef MF_TRACE
}

// Placement check, once per device (bevgen_finalize): the exchange inside ar_mlp_fused_kernel is only coherent if the workgroups i, i + 8, i + 16, ... of a launch share an
// XCD (one L2).  A 256-workgroup launch records HW_REG_XCC_ID per workgroup; any residue class that is split over two XCDs switches the fused launch off for this
// device (the kernel repeats the check on every launch and raises the context's error word, but a wrong placement should never get that far).
__global__ void xcc_probe_kernel(unsigned* out) {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    if (threadIdx.x == 0) out[blockIdx.x] = x & 0xf;
}
static bool xcd_placement_ok() {
    static std::atomic<int> state[MAX_DEVICES];   // 0 unknown, 1 ok, 2 not ok
    const int dev = current_device_slot();
    int st = state[dev].load(std::memory_order_acquire);
    if (st == 0) {
        unsigned* d = nullptr;
        unsigned h[256];
        bool ok = hipMalloc(reinterpret_cast<void**>(&d), sizeof h) == hipSuccess;
        if (ok) {
            hipLaunchKernelGGL(xcc_probe_kernel, dim3(256), dim3(512), 0, 0, d);
            ok = hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost) == hipSuccess;
            (void)hipFree(d);
        }
        for (int i = 8; ok && i < 256; ++i) ok = h[i] == h[i & 7];
        st = ok ? 1 : 2;
        state[dev].store(st, std::memory_order_release);
    }
    return st == 1;
}

bool xcd_placement_verified() { return xcd_placement_ok(); }

bool mlp_fused_supported(int M, int D, bool w_f16) {
    static const int env = getenv("BEVGEN_MLP_FUSE") ? atoi(getenv("BEVGEN_MLP_FUSE")) : 1;
    if (!env || M < 1 || M > 64 || D != 1024) return false;   // (the K slices per wave - 128 of D, 64 of the XCD's D / 2 - are written out for D = 1024)
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return false;
    (void)w_f16;
    return cus >= 4 * D / 16 && xcd_placement_ok();   // one workgroup per CU: every workgroup of an XCD must be resident at once; workgroup i on XCD i % 8
}

void launch_ar_mlp_fused(const MlpFusedArgs& g, hipStream_t s) {
    MlpFusedArgs ga = g;
    BG_REQUIRE(mlp_fused_supported(g.M, g.D, g.w_f16 != 0), "ar_mlp_fused: unsupported shape M=%d D=%d (or a device with fewer CUs than workgroups)", g.M, g.D);
    BG_REQUIRE(g.A && g.ln_w && g.ln_cs && g.ln_ds && g.Wup && g.Wdn && g.hidden && g.C && g.sync && g.err && g.lda % 4 == 0, "ar_mlp_fused: missing operand");
    static const int acq_env = getenv("BEVGEN_MLPF_ACQ") ? atoi(getenv("BEVGEN_MLPF_ACQ")) : 1;
    ga.acq = acq_env;
    const dim3 grid(4 * g.D / 16);
    // work = algorithmic bytes: both weight matrices once + rows in, hidden out and back, partial planes out
    ProfScope prof(PROF_GEMM_SKINNY, 8.0 * g.D * g.D * (g.w_f16 ? 2 : 4) + ((double)g.M * g.D + 2.0 * g.M * 4 * g.D + (double)MLP_FUSED_PLANES * g.M * g.D) * sizeof(float), s, true);
    if (g.w_f16) {
        if (prof.attached()) hipExtLaunchKernelGGL(ar_mlp_fused_kernel<1>, grid, dim3(SF_WAVES * 64), 0, s, prof.ev_a(), prof.ev_b(), 0, ga);
        else hipLaunchKernelGGL(ar_mlp_fused_kernel<1>, grid, dim3(SF_WAVES * 64), 0, s, ga);
    } else {
        if (prof.attached()) hipExtLaunchKernelGGL(ar_mlp_fused_kernel<0>, grid, dim3(SF_WAVES * 64), 0, s, prof.ev_a(), prof.ev_b(), 0, ga);
        else hipLaunchKernelGGL(ar_mlp_fused_kernel<0>, grid, dim3(SF_WAVES * 64), 0, s, ga);
    }
    LAUNCH_CHECK();
}

// W [N, K] row-major -> tile-major operand image: [N/16 column tiles][K/16 k-chunks][64 lanes][4]; lane l = r + 16 q holds W[16 tile + r][16 chunk + 4 q .. + 3],
// i.e. exactly the float4 that lane feeds to the four MFMAs of the chunk.  Rows beyond N are zero.
__global__ __launch_bounds__(256) void pack_skinny_weight_kernel(const float* __restrict__ W, float* __restrict__ Wp, int N, int K) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;   // one float4 of the packed image
    const long total = (long)((N + 15) / 16) * (K >> 4) * 64;
    if (i >= total) return;
    const int l = (int)(i & 63);
    const long ck = i >> 6;
    const int kc = (int)(ck % (K >> 4));
    const int tile = (int)(ck / (K >> 4));
    const int n = tile * 16 + (l & 15), k = kc * 16 + 4 * (l >> 4);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (n < N) v = *reinterpret_cast<const float4*>(W + (long)n * K + k);
    reinterpret_cast<float4*>(Wp)[i] = v;
}
// fp16 image: [N/16 column tiles][K/32 k-chunks][64 lanes][8 halves]; lane l = r + 16 q holds W[16 tile + r][32 chunk + 8 q .. + 7] rounded to fp16
__global__ __launch_bounds__(256) void pack_skinny_weight_f16_kernel(const float* __restrict__ W, _Float16* __restrict__ Wp, int N, int K, unsigned* __restrict__ status) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;   // one 8-half group of the packed image
    const long total = (long)((N + 15) / 16) * (K >> 5) * 64;
    if (i >= total) return;
    const int l = (int)(i & 63);
    const long ck = i >> 6;
    const int kc = (int)(ck % (K >> 5));
    const int tile = (int)(ck / (K >> 5));
    const int n = tile * 16 + (l & 15), k = kc * 32 + 8 * (l >> 4);
    half8_t v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = n < N ? (_Float16)W[(long)n * K + k + e] : (_Float16)0.f;
    unsigned bad = 0;
    guard_half8(v, bad);
    if (bad) status_raise(status, BG_ST_F16_RANGE);
    reinterpret_cast<half8_t*>(Wp)[i] = v;
}
void launch_pack_skinny_weight_f16(const float* W, void* Wp, int N, int K, hipStream_t s) {
    BG_REQUIRE(K % 32 == 0, "pack_skinny_weight_f16: K=%d must be a multiple of 32", K);
    const long total = (long)cdiv(N, 16) * (K >> 5) * 64;
    hipLaunchKernelGGL(pack_skinny_weight_f16_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, W, reinterpret_cast<_Float16*>(Wp), N, K, status_current());
    LAUNCH_CHECK();
}
size_t skinny_packed_floats(int N, int K) { return (size_t)cdiv(N, 16) * 16 * K; }
void launch_pack_skinny_weight(const float* W, float* Wp, int N, int K, hipStream_t s) {
    BG_REQUIRE(K % 16 == 0, "pack_skinny_weight: K=%d must be a multiple of 16", K);
    const long total = (long)cdiv(N, 16) * (K >> 4) * 64;
    hipLaunchKernelGGL(pack_skinny_weight_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, W, Wp, N, K);
    LAUNCH_CHECK();
}

int skinny_fused_ksplit(int N, int K) {
    // the K slice of one workgroup is at most 1024 (A tile in LDS, W slice in registers); beyond that, split until the grid covers the chip
    int s = 1;
    while (K / s > 1024 || (s < ROWSRC_RS_SPLITS && cdiv(N, 16) * s < 192 && (K / (s * 2)) % (SF_WAVES * 16) == 0 && K / (s * 2) >= 256)) s *= 2;
    return s;
}

bool skinny_fused_supported(int M, int N, int K, bool ln) {
    if (M < 1 || M > 64 || K % 4 != 0) return false;
    const int s = ln ? 1 : skinny_fused_ksplit(N, K);
    if (K % s != 0 || s > ROWSRC_RS_SPLITS) return false;   // the consumer's row source adds at most ROWSRC_RS_SPLITS partials
    const int kw = K / s;
    return kw <= 1024 && kw % (SF_WAVES * 16) == 0;
}

// fp16 weight images: the K slice of a workgroup must be a multiple of 32 per wave
bool skinny_fused_f16_ok(int N, int K, bool ln) {
    const int s = ln ? 1 : skinny_fused_ksplit(N, K);
    return K % s == 0 && (K / s) % (SF_WAVES * 32) == 0;
}

void launch_skinny_fused(const SkinnyFusedArgs& g0, hipStream_t s) {
    SkinnyFusedArgs g = g0;
    const bool ln = g.ln_w != nullptr;
    if (g.ksplit <= 0) g.ksplit = ln ? 1 : skinny_fused_ksplit(g.N, g.K);
    g.has_ln_b = g.ln_b != nullptr;
    if (ln && !g.ln_b) g.ln_b = g.ln_w;
    BG_REQUIRE(skinny_fused_supported(g.M, g.N, g.K, ln) && g.K % g.ksplit == 0 && (g.K / g.ksplit) <= 1024 && (g.K / g.ksplit) % (SF_WAVES * 16) == 0,
               "skinny_fused: unsupported shape M=%d N=%d K=%d ksplit=%d", g.M, g.N, g.K, g.ksplit);
    BG_REQUIRE(!ln || g.ksplit == 1, "skinny_fused: LayerNorm needs the whole row in one workgroup");
    BG_REQUIRE(g.ksplit <= ROWSRC_RS_SPLITS, "skinny_fused: at most %d K splits", ROWSRC_RS_SPLITS);
    BG_REQUIRE(g.lda % 4 == 0 && g.Wp, "skinny_fused: A stride must be a multiple of 4, weights packed");
    BG_REQUIRE(!g.a_src || (ln && g.a_src->ns <= ROWSRC_RS_SPLITS), "skinny_fused: a row source needs the LayerNorm form and at most %d partial sums", ROWSRC_RS_SPLITS);
    BG_REQUIRE(!g.xn_out || (ln && g.a_src), "skinny_fused: xn_out is the LayerNorm output of the row-source form");
    if (g.a_src) { g.src = rowsrc_fix(*g.a_src); g.a_src = nullptr; }
    const bool rs = g.src.base != nullptr;
    BG_REQUIRE(rs || g.A, "skinny_fused: no A operand");
    const bool fd = g.ln_cs != nullptr;
    BG_REQUIRE(!fd || (ln && !rs && g.ln_ds && g.K <= 1024), "skinny_fused: the folded LayerNorm needs the plain-A LayerNorm form and both row constants");
    dim3 grid(cdiv(g.N, 16), g.ksplit);
    if (g.w_f16) BG_REQUIRE((g.K / g.ksplit) % (SF_WAVES * 32) == 0, "skinny_fused: fp16 weights need a K slice that is a multiple of %d (K=%d, ksplit=%d)", SF_WAVES * 32, g.K, g.ksplit);
    ProfScope prof(PROF_GEMM_SKINNY, (double)g.N * g.K * (g.w_f16 ? 2 : 4) + ((double)g.M * g.K + (double)g.M * g.N) * sizeof(float), s, !rs || g.M <= 16);   // work = algorithmic bytes; one launch: events attached to it
#define SF_LAUNCH(K, ARGS) do { if (prof.attached()) hipExtLaunchKernelGGL(K, grid, dim3(SF_WAVES * 64), 0, s, prof.ev_a(), prof.ev_b(), 0, ARGS); \
                                else hipLaunchKernelGGL(K, grid, dim3(SF_WAVES * 64), 0, s, ARGS); } while (0)
    if (rs) {   // 16 rows per launch
        const int M = g.M;
        for (int m0 = 0; m0 < M; m0 += 16) {
            SkinnyFusedArgs h = g;
            h.M = std::min(16, M - m0);
            h.src.base += (long)m0 * g.src.ld;
            h.src.partial += (long)m0 * g.src.pld;   // (aliases base when there are no partials: any valid address will do)
            h.C = g.C + (long)m0 * g.ldc;
            if (g.xn_out) h.xn_out = g.xn_out + (long)m0 * g.ldxn;
            if (g.w_f16) SF_LAUNCH((skinny_fused_kernel<true, 1, true>), h);
            else SF_LAUNCH((skinny_fused_kernel<true, 0, true>), h);
        }
    } else if (g.w_f16) {
        if (ln && fd) SF_LAUNCH((skinny_fused_kernel<true, 1, false, true>), g);
        else if (ln) SF_LAUNCH((skinny_fused_kernel<true, 1>), g);
        else SF_LAUNCH((skinny_fused_kernel<false, 1>), g);
    } else {
        if (ln && fd) SF_LAUNCH((skinny_fused_kernel<true, 0, false, true>), g);
        else if (ln) SF_LAUNCH((skinny_fused_kernel<true, 0>), g);
        else SF_LAUNCH((skinny_fused_kernel<false, 0>), g);
    }
#undef SF_LAUNCH
    LAUNCH_CHECK();
}

}  // namespace bevgen
