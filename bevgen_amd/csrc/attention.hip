// Attention kernels (head dim 64) for gfx950.
//
// 1. attention_fwd_kernel - flash-style (online softmax) attention on the fp32 matrix cores, used for
//      * Route M self / cross attention with the additive camera-bias matrix and the null key   (muse_net:148-166)
//      * Route A prefill over the BEV condition rows (dense restatement of ssa:150-176)
//    O = softmax(scale * Q K^T + bias) V, bias rows indexed by query, masks baked into the bias as -1e30.
//
//    Wave layout (64 lanes, v_mfma_f32_32x32x2_f32): a wave owns 32 query rows and walks 32-key tiles.
//    Scores are computed TRANSPOSED, S^T = K Q^T, so the MFMA result puts the query index on the lane axis
//    (col = lane&31) and 16 keys on the register axis: the row-softmax is in-lane work plus ONE xor-32 exchange,
//    and exp(S^T) is already laid out as the B operand of the second MFMA, O^T = V^T P^T - no LDS round trip for P.
//    The lane halves (lane>>5) own disjoint k-slots of each MFMA; we give half h the contiguous head-dim range
//    [32h, 32h+32) for QK^T (one ds_read_b128 of K feeds 4 MFMAs) and keys {(r&3)+8(r>>2)+4h} for PV, which is
//    exactly the key set its score registers hold.
//
// 2. decode_attention_kernel (+ combine) - Route A per-token decode against the growing KV cache: one query row per
//    (sequence, head), HBM-bandwidth bound.  Lanes are laid out so every vector load is a fully coalesced 1 KiB
//    (LPK lanes x 16 B cover one key row; 64/LPK keys per instruction); split-K over the context for occupancy.
#include "common.h"
#include "kernels.h"
#include "profiler.h"
#include <hip/hip_bf16.h>

namespace bevgen {

constexpr int KT = 32;        // keys per tile
constexpr int KLD = 68;       // padded K row stride in LDS (floats): 16-lane groups of ds_read_b128 -> distinct bank quads
constexpr int VLD = 64;

__global__ __launch_bounds__(256, 2) void attention_fwd_kernel(AttnArgs a) {
    __shared__ __attribute__((aligned(16))) float Ks[2][KT * KLD];
    __shared__ __attribute__((aligned(16))) float Vs[2][KT * VLD];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int qi = lane & 31, h = lane >> 5;
    const int head = blockIdx.y, b = blockIdx.z;
    const int q0 = blockIdx.x * 128 + wave * 32;
    const int qrow = q0 + qi;
    const bool qvalid = qrow < a.Nq;
    const int qclamped = qvalid ? qrow : a.Nq - 1;

    const float* Qp = a.Q + (long)b * a.q_bstride + (long)head * a.q_hstride + (long)qclamped * 64 + 32 * h;
    const float* Kp = a.K + (long)(b / a.kv_group) * a.kv_bstride + (long)head * a.kv_hstride;
    const float* Vp = a.V + (long)(b / a.kv_group) * a.kv_bstride + (long)head * a.kv_hstride;
    const float* Bp = a.bias ? a.bias + (long)head * a.bias_head_stride + (long)qclamped * a.ldbias + 4 * h : nullptr;

    float qf[32];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const float4 v = *reinterpret_cast<const float4*>(Qp + 4 * c);
        qf[4 * c + 0] = v.x; qf[4 * c + 1] = v.y; qf[4 * c + 2] = v.z; qf[4 * c + 3] = v.w;
    }

    f32x16 accO[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) accO[t][r] = 0.f;
    float m_run = kNegBig, l_run = 0.f;

    // tile loader: 32 rows x 16 float4 = 512 float4 per matrix -> 2 per thread
    const int lrow0 = tid >> 4, lc4 = tid & 15;  // rows lrow0 and lrow0+16
    float4 rk0, rk1, rv0, rv1;  // named registers (arrays captured by the lambdas below would be demoted to scratch)
    auto gload = [&](int tile) {
        const long off0 = (long)(tile * KT + lrow0) * 64 + lc4 * 4, off1 = off0 + 16 * 64;
        rk0 = *reinterpret_cast<const float4*>(Kp + off0);
        rv0 = *reinterpret_cast<const float4*>(Vp + off0);
        rk1 = *reinterpret_cast<const float4*>(Kp + off1);
        rv1 = *reinterpret_cast<const float4*>(Vp + off1);
    };
    auto lstore = [&](int buf) {
        *reinterpret_cast<float4*>(&Ks[buf][lrow0 * KLD + lc4 * 4]) = rk0;
        *reinterpret_cast<float4*>(&Vs[buf][lrow0 * VLD + lc4 * 4]) = rv0;
        *reinterpret_cast<float4*>(&Ks[buf][(lrow0 + 16) * KLD + lc4 * 4]) = rk1;
        *reinterpret_cast<float4*>(&Vs[buf][(lrow0 + 16) * VLD + lc4 * 4]) = rv1;
    };

    // key tiles to visit: all of them, or the list of this (head, query block) - tiles no row of the block can see are neither loaded nor multiplied
    const uint16_t* tl = a.tiles ? a.tiles + (long)head * a.tiles_head_stride + (long)blockIdx.x * a.tiles_ld : nullptr;
    const int ntiles = tl ? (int)tl[0] : a.Nk_pad / KT;
    auto tile_at = [&](int i) { return tl ? (int)tl[1 + i] : i; };
    if (ntiles > 0) {
        gload(tile_at(0));
        lstore(0);
    }
    __syncthreads();
    int cur = 0;
    for (int ti = 0; ti < ntiles; ++ti) {
        const int tile = tile_at(ti);
        const bool more = ti + 1 < ntiles;
        if (more) gload(tile_at(ti + 1));

        // ---- S^T = K Q^T  (A = K rows from LDS, B = Q registers)
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
        const float* ks = &Ks[cur][qi * KLD + 32 * h];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float4 kf = *reinterpret_cast<const float4*>(ks + 4 * c);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.x, qf[4 * c + 0], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.y, qf[4 * c + 1], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.z, qf[4 * c + 2], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.w, qf[4 * c + 3], s, 0, 0, 0);
        }
        // s[r] = S^T[key = (r&3) + 8(r>>2) + 4h][q = lane&31]

        // ---- scale + bias, online softmax (per-lane query row; the two halves hold disjoint keys of the same row)
        float mx = kNegBig;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (Bp) bv = *reinterpret_cast<const float4*>(Bp + tile * KT + 8 * g);
            s[4 * g + 0] = s[4 * g + 0] * a.scale + bv.x;
            s[4 * g + 1] = s[4 * g + 1] * a.scale + bv.y;
            s[4 * g + 2] = s[4 * g + 2] * a.scale + bv.z;
            s[4 * g + 3] = s[4 * g + 3] * a.scale + bv.w;
            mx = fmaxf(mx, fmaxf(fmaxf(s[4 * g], s[4 * g + 1]), fmaxf(s[4 * g + 2], s[4 * g + 3])));
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = expf(m_run - m_new);
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[r] = expf(s[r] - m_new);
            psum += s[r];
        }
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) accO[t][r] *= alpha;

        // ---- O^T += V^T P^T  (A = V^T: lane = head-dim column, B = P registers)
        const float* vs = &Vs[cur][(4 * h) * VLD + qi];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = (r & 3) + 8 * (r >> 2);
            const float v0 = vs[key * VLD];
            const float v1 = vs[key * VLD + 32];
            accO[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(v0, s[r], accO[0], 0, 0, 0);
            accO[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(v1, s[r], accO[1], 0, 0, 0);
        }

        if (more) lstore(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue: accO[t][r] = O^T[d = 32t + (r&3) + 8(r>>2) + 4h][q]
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.f / l_tot;
    if (qvalid) {
        const long orow = (long)b * a.o_bstride + (long)qrow * a.o_qstride + (long)head * a.o_hstride;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d = 32 * t + 8 * g + 4 * h;
                float4 o = make_float4(accO[t][4 * g] * inv, accO[t][4 * g + 1] * inv, accO[t][4 * g + 2] * inv, accO[t][4 * g + 3] * inv);
                if (a.R) {
                    const float4 rr = *reinterpret_cast<const float4*>(a.R + orow + d);
                    o.x += rr.x; o.y += rr.y; o.z += rr.z; o.w += rr.w;
                }
                *reinterpret_cast<float4*>(a.O + orow + d) = o;
            }
    }
}

// tiles[h][qb] = {count, ids...}: key tile t is listed when any row of query block qb (128 rows) has a bias value above the mask level in it
__global__ __launch_bounds__(256) void build_attn_tiles_kernel(const float* __restrict__ bias, long bias_head_stride, int ldbias, int Nq, int Nk_pad, uint16_t* __restrict__ tiles,
                                                               int tiles_ld) {
    __shared__ int present[1024];
    const int qb = blockIdx.x, h = blockIdx.y, nt = Nk_pad / KT;
    for (int t = threadIdx.x; t < nt; t += blockDim.x) present[t] = 0;
    __syncthreads();
    const float* B = bias + (long)h * bias_head_stride;
    const int r1 = min(Nq, qb * 128 + 128);
    for (int r = qb * 128; r < r1; ++r)
        for (int c = threadIdx.x; c < Nk_pad; c += blockDim.x)
            if (B[(long)r * ldbias + c] > 0.5f * kNegBig) present[c / KT] = 1;   // (benign race: every writer stores 1)
    __syncthreads();
    if (threadIdx.x == 0) {
        uint16_t* out = tiles + ((long)h * gridDim.x + qb) * tiles_ld;
        int n = 0;
        for (int t = 0; t < nt; ++t)
            if (present[t]) out[1 + n++] = (uint16_t)t;
        out[0] = (uint16_t)n;
    }
}
size_t attn_tiles_elems(int heads, int Nq, int Nk_pad) { return (size_t)heads * cdiv(Nq, 128) * (Nk_pad / KT + 1); }
void launch_build_attn_tiles(const float* masked_bias, long bias_head_stride, int ldbias, int heads, int Nq, int Nk_pad, uint16_t* tiles, hipStream_t s) {
    BG_REQUIRE(Nk_pad / KT <= 1024, "attention tile lists: at most 1024 key tiles");
    hipLaunchKernelGGL(build_attn_tiles_kernel, dim3(cdiv(Nq, 128), heads), dim3(256), 0, s, masked_bias, bias_head_stride, ldbias, Nq, Nk_pad, tiles, Nk_pad / KT + 1);
    LAUNCH_CHECK();
}

void launch_attention(const AttnArgs& a, hipStream_t s) {
    BG_REQUIRE(a.Nk_pad % KT == 0 && a.Nk_pad > 0, "attention: Nk_pad=%d must be a positive multiple of %d", a.Nk_pad, KT);
    BG_REQUIRE(a.bias == nullptr || a.ldbias % 4 == 0, "attention: bias row stride must be a multiple of 4");
    dim3 grid(cdiv(a.Nq, 128), a.H, a.B);
    ProfScope prof(PROF_ATTN, 4.0 * a.B * a.H * (double)a.Nq * a.Nk_pad * 64, s);
    hipLaunchKernelGGL(attention_fwd_kernel, grid, dim3(256), 0, s, a);
    LAUNCH_CHECK();
}

// =====================================================================================================
// Decode attention
// =====================================================================================================
template <int DT> struct KvTraits;
template <> struct KvTraits<0> {  // fp32 rows: 256 B = 16 lanes x 16 B
    static constexpr int LPK = 16, DPL = 4;
    __device__ static __forceinline__ void load(const void* base, long row, int sub, float (&out)[4]) {
        const float4 v = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + row * 64 + sub * 4);
        out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
    }
};
template <> struct KvTraits<1> {  // fp16 rows: 128 B = 8 lanes x 16 B
    static constexpr int LPK = 8, DPL = 8;
    __device__ static __forceinline__ void load(const void* base, long row, int sub, float (&out)[8]) {
        typedef _Float16 h8 __attribute__((ext_vector_type(8)));
        const h8 v = *reinterpret_cast<const h8*>(reinterpret_cast<const _Float16*>(base) + row * 64 + sub * 8);
#pragma unroll
        for (int i = 0; i < 8; ++i) out[i] = (float)v[i];
    }
};

constexpr int DEC_UNROLL = 4;

// partial results: ws[((b*H + h)*S + split)*66 + {0: m, 1: l, 2..65: o[64]}]
// NW = waves per workgroup.  S == 1 (one workgroup sees the whole context): the epilogue normalises, adds the residual and writes O
// directly - no workspace round trip, no second kernel.  S > 1: partials go to `ws` and decode_attention_combine_kernel merges them.
constexpr float kLog2eDec = 1.44269504088896340736f;

// all-reduce (sum) inside aligned groups of LPK = 8 or 16 lanes with DPP row operations (VALU speed) instead of ds_bpermute round trips through
// the LDS crossbar: quad_perm xor 1, quad_perm xor 2, row_half_mirror (lane i <-> 7-i of its 8), row_mirror (i <-> 15-i of its 16) - the mirror steps
// are valid for a symmetric reduction because every lane of a quad already holds the quad's sum
template <int LPK>
__device__ __forceinline__ float group_sum(float d) {
    d += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(d), 0xB1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
    d += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(d), 0x4E, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
    d += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(d), 0x141, 0xf, 0xf, true));   // row_half_mirror
    if (LPK == 16) d += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(d), 0x140, 0xf, 0xf, true));   // row_mirror
    return d;
}

template <int DT, int NW>
__global__ __launch_bounds__(NW * 64) void decode_attention_kernel(DecodeAttnArgs a, float* __restrict__ ws, int S) {
    using T = KvTraits<DT>;
    constexpr int LPK = T::LPK, DPL = T::DPL, KPI = 64 / LPK;
    __shared__ float red[NW][66];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sub = lane % LPK, kslot = lane / LPK;
    const int split = blockIdx.x, head = blockIdx.y, b = blockIdx.z;
    // context length: frozen argument, or (graph replay) read from the device-side step counter
    const int n = a.d_n ? *a.d_n + a.n : a.n;
    const int row = n - 1;  // sequence row of the query (it attends to keys 0..n-1, itself included)
    const int chunk = ((n + S - 1) / S + 63) & ~63;
    const int k_begin = split * chunk;
    const int k_end = min(n, k_begin + chunk);

    float qv[DPL];
    {
        const float* qp = a.q + (long)b * a.ldq + head * 64 + sub * DPL;
#pragma unroll
        for (int i = 0; i < DPL; ++i) qv[i] = qp[i] * (a.scale * kLog2eDec);   // scores live in the base-2 domain: exp(x) = 2^(x log2 e), one v_exp_f32
    }
    const long cache_row0 = ((long)b * a.H + head) * a.Lmax;
    // visibility of key k = element mask x block layout (SparseVis); this per-operator kernel masks per key and walks every key
    const SparseVis& vis = a.vis;
    const uint8_t* arow = vis.allowed + (long)head * vis.allowed_head_stride + (long)row * vis.ldallowed;
    const uint8_t* lrow = vis.lay + (long)head * vis.lay_head_stride + (long)(row / vis.blk) * vis.nb;
    const float* bias_row = a.bias ? a.bias + (long)row * a.ldbias : nullptr;

    // fused KV append: the workgroup that owns the newest key writes this step's k/v row into the cache, then everybody reads it back
    // through the normal path (same-workgroup visibility is guaranteed by the barrier)
    if (a.append_k && k_end == n) {
        if (tid < 128) {
            const bool is_v = tid >= 64;
            const int d = tid & 63;
            const float val = (is_v ? a.append_v : a.append_k)[(long)b * a.ldq + head * 64 + d];
            void* cache = const_cast<void*>(is_v ? a.vcache : a.kcache);
            const long idx = (cache_row0 + row) * 64 + d;
            if (DT == 0) reinterpret_cast<float*>(cache)[idx] = val;
            else {
                const _Float16 hv = (_Float16)val;
                unsigned bad = 0;
                guard_half(hv, bad);
                if (bad) status_raise(a.status, BG_ST_F16_RANGE);
                reinterpret_cast<_Float16*>(cache)[idx] = hv;
            }
        }
        __syncthreads();
    }

    float m_run = kNegBig, l_run = 0.f, acc[DPL];
#pragma unroll
    for (int i = 0; i < DPL; ++i) acc[i] = 0.f;

    constexpr int stride = NW * KPI;  // keys consumed per unroll slot by the workgroup's waves
    const int iters = k_end > k_begin ? (k_end - k_begin + stride * DEC_UNROLL - 1) / (stride * DEC_UNROLL) : 0;
    for (int it = 0; it < iters; ++it) {
        float kx[DEC_UNROLL][DPL], vx[DEC_UNROLL][DPL], sc[DEC_UNROLL];
        int key[DEC_UNROLL];
#pragma unroll
        for (int u = 0; u < DEC_UNROLL; ++u) {
            key[u] = k_begin + (it * DEC_UNROLL + u) * stride + wave * KPI + kslot;
            const int kc = min(key[u], k_end - 1);  // clamped address, masked below
            T::load(a.kcache, cache_row0 + kc, sub, kx[u]);
            T::load(a.vcache, cache_row0 + kc, sub, vx[u]);
        }
        float mx = m_run;
#pragma unroll
        for (int u = 0; u < DEC_UNROLL; ++u) {
            float d = 0.f;
#pragma unroll
            for (int i = 0; i < DPL; ++i) d = fmaf(qv[i], kx[u][i], d);
            d = group_sum<LPK>(d);
            const int kc = min(key[u], k_end - 1);
            const bool ok = key[u] < k_end && (!vis.has_allowed || arow[kc]) && (!vis.has_lay || lrow[kc / vis.blk]);
            sc[u] = ok ? d + (bias_row ? bias_row[kc] * (a.scale * kLog2eDec) : 0.f) : kNegBig;
            mx = fmaxf(mx, sc[u]);
        }
        const float alpha = __builtin_amdgcn_exp2f(m_run - mx);
        l_run *= alpha;
#pragma unroll
        for (int i = 0; i < DPL; ++i) acc[i] *= alpha;
#pragma unroll
        for (int u = 0; u < DEC_UNROLL; ++u) {
            const float p = sc[u] <= kNegBig ? 0.f : __builtin_amdgcn_exp2f(sc[u] - mx);
            l_run += p;
#pragma unroll
            for (int i = 0; i < DPL; ++i) acc[i] = fmaf(p, vx[u][i], acc[i]);
        }
        m_run = mx;
    }

    // ---- combine the key slots of this wave (lanes with equal `sub`); l_run is identical in the LPK lanes of a slot
    float m_all = m_run;
#pragma unroll
    for (int o = LPK; o < 64; o <<= 1) m_all = fmaxf(m_all, __shfl_xor(m_all, o, 64));
    const float f = __builtin_amdgcn_exp2f(m_run - m_all);
    l_run *= f;
#pragma unroll
    for (int i = 0; i < DPL; ++i) acc[i] *= f;
#pragma unroll
    for (int o = LPK; o < 64; o <<= 1) {
        l_run += __shfl_xor(l_run, o, 64);
#pragma unroll
        for (int i = 0; i < DPL; ++i) acc[i] += __shfl_xor(acc[i], o, 64);
    }
    if (kslot == 0) {
        if (sub == 0) { red[wave][0] = m_all; red[wave][1] = l_run; }
#pragma unroll
        for (int i = 0; i < DPL; ++i) red[wave][2 + sub * DPL + i] = acc[i];
    }
    __syncthreads();
    if (tid < 64) {
        float mm = kNegBig;
#pragma unroll
        for (int w = 0; w < NW; ++w) mm = fmaxf(mm, red[w][0]);
        float l = 0.f, o = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const float fw = __builtin_amdgcn_exp2f(red[w][0] - mm);
            l += red[w][1] * fw;
            o += red[w][2 + tid] * fw;
        }
        if (S == 1) {
            float v = o / l;
            if (a.R) v += a.R[(long)b * a.ldr + head * 64 + tid];
            a.O[(long)b * a.ldo + head * 64 + tid] = v;
        } else {
            float* out = ws + (((long)b * a.H + head) * S + split) * 66;
            if (tid == 0) { out[0] = mm; out[1] = l; }
            out[2 + tid] = o;
        }
    }
}

__global__ __launch_bounds__(64) void decode_attention_combine_kernel(DecodeAttnArgs a, const float* __restrict__ ws, int S) {
    const int head = blockIdx.x, b = blockIdx.y, d = threadIdx.x;
    const float* p = ws + (((long)b * a.H + head) * S) * 66;
    float mm = kNegBig;
    for (int s = 0; s < S; ++s) mm = fmaxf(mm, p[s * 66]);
    float l = 0.f, o = 0.f;
    for (int s = 0; s < S; ++s) {
        const float f = __builtin_amdgcn_exp2f(p[s * 66] - mm);   // partial maxima are in the base-2 domain (decode_attention_kernel)
        l += p[s * 66 + 1] * f;
        o += p[s * 66 + 2 + d] * f;
    }
    float v = o / l;
    if (a.R) v += a.R[(long)b * a.ldr + head * 64 + d];
    a.O[(long)b * a.ldo + head * 64 + d] = v;
}

void launch_decode_attention_combine(const DecodeAttnArgs& a, const float* ws, int S, hipStream_t s) {
    hipLaunchKernelGGL(decode_attention_combine_kernel, dim3(a.H, a.B), dim3(64), 0, s, a, ws, S);
    LAUNCH_CHECK();
}

int decode_attention_splits(int B, int H, int n_max) {
    // B*H >= 192 workgroups already cover the chip: one 16-wave workgroup per (sequence, head), direct epilogue.
    // Fewer than that: split the context over workgroups (at least 256 keys per split) and merge in a second kernel.
    if ((long)B * H >= 192) return 1;
    int S = 1;
    while ((long)B * H * S < 1024 && n_max / (S * 2) >= 256) S *= 2;
    return S;
}

size_t decode_attention_ws_bytes(int B, int H, int S) { return (size_t)B * H * S * 66 * sizeof(float); }

void launch_decode_attention_ws(const DecodeAttnArgs& a0, float* ws, int S, hipStream_t s) {
    DecodeAttnArgs a = a0;
    a.vis = vis_fix(a.vis, a.q);
    a.status = status_current();
    BG_REQUIRE(a.d_n || (a.n > 0 && a.n <= a.Lmax), "decode attention: n=%d out of range (Lmax=%d)", a.n, a.Lmax);
    BG_REQUIRE(a.group <= 1, "decode attention: shared-prefix groups not implemented in this kernel");
    dim3 grid(S, a.H, a.B);
    // algorithmic bytes of one launch: K and V rows of the visible context, once each (2 * n * 64 * elem per (sequence, head))
    const double n_host = a.d_n ? a.n + a.n_hint : a.n;
    ProfScope prof(PROF_DECODE_ATTN, 2.0 * a.B * a.H * n_host * 64 * (a.kv_dtype == 0 ? 4 : 2), s);
    if (S == 1) {
        if (a.kv_dtype == 0)
            hipLaunchKernelGGL((decode_attention_kernel<0, 16>), grid, dim3(1024), 0, s, a, ws, S);
        else
            hipLaunchKernelGGL((decode_attention_kernel<1, 16>), grid, dim3(1024), 0, s, a, ws, S);
        LAUNCH_CHECK();
        return;
    }
    if (a.kv_dtype == 0)
        hipLaunchKernelGGL((decode_attention_kernel<0, 4>), grid, dim3(256), 0, s, a, ws, S);
    else
        hipLaunchKernelGGL((decode_attention_kernel<1, 4>), grid, dim3(256), 0, s, a, ws, S);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(decode_attention_combine_kernel, dim3(a.H, a.B), dim3(64), 0, s, a, ws, S);
    LAUNCH_CHECK();
}

}  // namespace bevgen
