// Split-precision flash attention (head dim 64) for Route M:  O = softmax(scale * Q K^T + bias) V  (scale and bias arrive multiplied by log2 e and
// the exponentials are v_exp_f32 = 2^x: one instruction instead of the library expf's twelve in a VALU-bound loop) with fp32-class accuracy on the f16
// matrix cores.  Same structure as attention.hip's fp32 kernel (scores computed transposed so the query index sits on the lane axis, the
// softmax is in-lane plus one xor-32 exchange, exp(S^T) is already the B operand of O^T = V^T P^T), but every matrix product is evaluated as
//        X Y^T ~= hi_x hi_y^T + 2^-11 (hi_x lo_y^T + lo_x hi_y^T),      x = hi + lo * 2^-11 (two f16 numbers, 22 mantissa bits)
// with THREE v_mfma_f32_32x32x16_f16 per 16-deep k-step instead of EIGHT v_mfma_f32_32x32x2_f32 (see gemm_split.hip).
//   * Q, K arrive pre-split from the q/k preparation kernels (embed.hip: l2norm + scale + split), V pre-split AND transposed
//     ([B,H,64,Nk_pad]) so that the V^T operand rows are contiguous along the key axis;
//   * P = exp(S - m) is split in registers; the score registers of lane half h hold keys {(r&3) + 8(r>>2) + 4h}, so for the k-step that
//     covers keys [16s, 16s+16) half h owns keys 16s + {0..3, 8..11} + 4h - the V^T fragment is gathered with the same permutation
//     (two 8-byte LDS reads), which is legal because any k permutation applied to both MFMA operands leaves the product unchanged;
//   * LDS rows are padded (K: 72 halves, V^T: 36 halves) so that ds_read_b128 / ds_read_b64 lane groups are bank-conflict free.
#include "common.h"
#include "kernels.h"
#include "profiler.h"

namespace bevgen {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

constexpr int SKT = 32;    // keys per tile
constexpr int SKLD = 72;   // K tile row stride (halves)
constexpr int SVLD = 36;   // V^T tile row stride (halves)
constexpr float kLo = 2048.f, kLoI = 1.f / 2048.f;

// Variant bits (the production launcher uses kAttnVariant; tools/attn_lab times the others against it):
//   AV_BIASACC  the bias row segment is the C operand of the first QK^T MFMA (its register layout IS the accumulator layout) instead of 16 adds
//   AV_MIX      P_lo = fma_mix(p - hi) straight to f16 (v_fma_mixlo/hi_f16) instead of cvt + sub + cvt
//   AV_PRIO     s_setprio 1 around the MFMA blocks
//   AV_NOSOFT / AV_NOPV / AV_NOQK   diagnostics only (wrong results): drop the softmax VALU work / the PV MFMAs / the QK MFMAs
//   AV_BIASPK   the bias arrives as a packed image [q block of 32][key tile][4][64 lanes][4]: every load instruction of a wave reads 1 KiB contiguous
//   AV_VTILE    V^T planes are tile-major [key tile][64 dims][32 keys]: a tile is 4 KiB contiguous (was 64-byte pieces at a row stride of 2 Nk_pad bytes)
enum { AV_BIASACC = 1, AV_MIX = 2, AV_PRIO = 4, AV_BIASPK = 8, AV_VTILE = 16, AV_NOSOFT = 32, AV_NOPV = 64, AV_NOQK = 128, AV_ONEACC = 256, AV_BUNCH = 2048, AV_NOPRELOAD = 4096, AV_LOCKSTEP = 8192, AV_VPRIO = 16384, AV_PKF32 = 32768, AV_SPLITM = 65536, AV_TRACE = 1 << 20 };
#ifdef BEVGEN_ATTN_LAB
__device__ unsigned long long g_attn_trace[8 * 16 * 8];   // [wave][tile][stamp]
#define PP_STAMP(k) do { if ((VAR & AV_TRACE) && lane == 0 && t < 16) trace_lds[(wave * 16 + t) * 8 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define PP_STAMP(k) do { } while (0)
#endif
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int VAR>
__global__ __launch_bounds__(256, 2) void attention_split_kernel(AttnSplitArgs a) {
    __shared__ __attribute__((aligned(16))) _Float16 Kh[2][SKT * SKLD], Kl[2][SKT * SKLD];
    __shared__ __attribute__((aligned(16))) _Float16 Vh[2][64 * SVLD], Vl[2][64 * SVLD];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int qi = lane & 31, h = lane >> 5;
    const int head = blockIdx.y, b = blockIdx.z;
    const int qrow = blockIdx.x * 128 + wave * 32 + qi;
    const bool qvalid = qrow < a.Nq;
    const int qc = qvalid ? qrow : a.Nq - 1;

    const long qoff = ((long)b * a.H + head) * a.Nq * 64 + (long)qc * 64 + 8 * h;
    const long koff = ((long)b * a.H + head) * (long)a.Nk_pad * 64;
    // bias == nullptr: the launcher points it at a zero row with ldbias = 0 (unconditional loads; no select / branch in the loop)
    const float* Bp = (VAR & AV_BIASPK) ? a.bias_pk + (long)head * a.bias_head_stride + (long)(blockIdx.x * 4 + wave) * a.bias_pk_qb_stride + lane * 4
                                        : a.bias + (long)head * a.bias_head_stride + (long)qc * a.ldbias + 4 * h;
    const int bstep = (VAR & AV_BIASPK) ? a.bias_pk_tile_step : a.bias_tile_step;   // floats per key tile; 0 for the zero row

    half8 qh[4], ql[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        qh[s] = *reinterpret_cast<const half8*>(a.Qh + qoff + 16 * s);
        ql[s] = *reinterpret_cast<const half8*>(a.Ql + qoff + 16 * s);
    }

    f32x16 oM[2], oC[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) { oM[t][r] = 0.f; oC[t][r] = 0.f; }
    float m_run = kNegBig, l_run = 0.f;

    // tile loaders: K planes 32 rows x 128 B = 256 x 16 B chunks (1 per thread); V^T planes 64 rows x 64 B = 256 x 16 B chunks (1 per thread)
    const int kr = tid >> 3, kc = tid & 7;   // K: row, 16-byte chunk
    const int vr = tid >> 2, vc = tid & 3;   // V^T: row (head dim), 16-byte chunk (8 keys)
    uint4 rkh, rkl, rvh, rvl;
    auto gload = [&](int tile) {
        const long ko = koff + (long)(tile * SKT + kr) * 64 + kc * 8;
        rkh = *reinterpret_cast<const uint4*>(a.Kh + ko);
        rkl = *reinterpret_cast<const uint4*>(a.Kl + ko);
        // V^T [.., 64, Nk_pad] (or tile-major [.., Nk_pad/32, 64, 32]): same number of elements per (b,h) as K
        const long vo = (VAR & AV_VTILE) ? koff + (long)tile * (64 * SKT) + vr * SKT + vc * 8 : koff + (long)vr * a.Nk_pad + tile * SKT + vc * 8;
        rvh = *reinterpret_cast<const uint4*>(a.VTh + vo);
        rvl = *reinterpret_cast<const uint4*>(a.VTl + vo);
    };
    auto lstore = [&](int buf) {
        *reinterpret_cast<uint4*>(&Kh[buf][kr * SKLD + kc * 8]) = rkh;
        *reinterpret_cast<uint4*>(&Kl[buf][kr * SKLD + kc * 8]) = rkl;
        // 72-byte rows: 8-byte aligned only -> two 8-byte stores
        uint2* dh = reinterpret_cast<uint2*>(&Vh[buf][vr * SVLD + vc * 8]);
        uint2* dl = reinterpret_cast<uint2*>(&Vl[buf][vr * SVLD + vc * 8]);
        dh[0] = make_uint2(rvh.x, rvh.y); dh[1] = make_uint2(rvh.z, rvh.w);
        dl[0] = make_uint2(rvl.x, rvl.y); dl[1] = make_uint2(rvl.z, rvl.w);
    };

    const int ntiles = a.Nk_pad / SKT;
    // bias row segment of a key tile (this lane's 16 keys, in accumulator order), fetched one tile ahead: an L2 round trip is longer than the
    // QK^T MFMAs in front of it
    f32x16 bacc;
    auto gload_bias = [&](int tile) {
        const float* src = Bp + tile * bstep;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 v = *reinterpret_cast<const float4*>(src + ((VAR & AV_BIASPK) ? 256 * g : 8 * g));
            bacc[4 * g] = v.x; bacc[4 * g + 1] = v.y; bacc[4 * g + 2] = v.z; bacc[4 * g + 3] = v.w;
        }
    };
    gload_bias(0);
    gload(0);
    lstore(0);
    if (ntiles > 1) gload(1);
    __syncthreads();
    int cur = 0;
    for (int tile = 0; tile < ntiles; ++tile) {
        const bool more = tile + 1 < ntiles;
        // staging (guide T14, "write after the barrier"): the registers hold tile+1 (requested one iteration ago); it goes into the buffer the previous
        // iteration just finished reading, and the same registers are re-issued at once for tile+2 - the ds_writes no longer queue behind this
        // iteration's MFMAs in front of the barrier
        if (more) lstore(cur ^ 1);
        if (tile + 2 < ntiles) gload(tile + 2);

        // ---- S^T = K Q^T
        f32x16 sM, sC;  // main and correction accumulator (the two correction products of a k-step are chained on sC with the main MFMA between them)
        const _Float16* kh = &Kh[cur][qi * SKLD + 8 * h];
        const _Float16* kl = &Kl[cur][qi * SKLD + 8 * h];
        if (VAR & AV_PRIO) __builtin_amdgcn_s_setprio(1);
        if (!(VAR & AV_NOQK)) {
            {   // first k-step: accumulate onto the inline constant 0 (no 32 v_mov per tile to clear the accumulators) / onto the bias
                const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                const half8 ah = *reinterpret_cast<const half8*>(kh);
                const half8 al = *reinterpret_cast<const half8*>(kl);
                sC = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, ql[0], zero, 0, 0, 0);
                sM = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, qh[0], (VAR & AV_BIASACC) ? bacc : zero, 0, 0, 0);
                sC = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, qh[0], sC, 0, 0, 0);
            }
#pragma unroll
            for (int s = 1; s < 4; ++s) {
                const half8 ah = *reinterpret_cast<const half8*>(kh + 16 * s);
                const half8 al = *reinterpret_cast<const half8*>(kl + 16 * s);
                sC = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, ql[s], sC, 0, 0, 0);
                sM = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, qh[s], sM, 0, 0, 0);
                sC = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, qh[s], sC, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) { sM[r] = (float)kh[r] + bacc[r]; sC[r] = (float)kl[r]; }
        }
        if (VAR & AV_PRIO) __builtin_amdgcn_s_setprio(0);

        // ---- scale + bias, online softmax
        float sv[16];
        float mx = kNegBig;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            sv[r] = fmaf(sC[r], kLoI, sM[r]);   // the score scale lives in the Q planes
            if (!(VAR & AV_BIASACC)) sv[r] += bacc[r];
            mx = fmaxf(mx, sv[r]);
        }
        if (more) gload_bias(tile + 1);
        half8 ph[2], pl[2];
        if (!(VAR & AV_NOSOFT)) {
            mx = fmaxf(mx, xor32(mx));
            const float m_new = fmaxf(m_run, mx);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            float psum = 0.f;
            // p in [0,1] = hi + lo with lo stored UNSCALED: |lo| <= 2^-12, and what the f16 subnormal range drops is below 2^-25 ABSOLUTE, which is
            // what matters for a probability (a weight of the row sum); saves the 2^11 scaling and lets V_hi P_lo accumulate straight into oM
            if (VAR & AV_MIX) {
                // hi: one v_cvt_pk_f16_f32 per pair; lo = p - hi evaluated in fp32 and rounded to f16 by ONE v_fma_mix{lo,hi}_f16 per element (the f16
                // operand is read in place from the packed hi word) instead of v_cvt_f32_f16 + v_sub_f32 + half a v_cvt_pk
                uint32_t hw[8], lw[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const float p0 = __builtin_amdgcn_exp2f(sv[2 * r] - m_new), p1 = __builtin_amdgcn_exp2f(sv[2 * r + 1] - m_new);
                    psum += p0;
                    psum += p1;
                    const half2v hp = {(_Float16)p0, (_Float16)p1};
                    hw[r] = __builtin_bit_cast(uint32_t, hp);
                    uint32_t l;
                    asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(l) : "v"(p0), "v"(hw[r]));
                    asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(p1), "v"(hw[r]));
                    lw[r] = l;
                }
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    ph[s] = __builtin_bit_cast(half8, u32x4{hw[4 * s], hw[4 * s + 1], hw[4 * s + 2], hw[4 * s + 3]});
                    pl[s] = __builtin_bit_cast(half8, u32x4{lw[4 * s], lw[4 * s + 1], lw[4 * s + 2], lw[4 * s + 3]});
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float p = __builtin_amdgcn_exp2f(sv[r] - m_new);   // scores are in the base-2 domain (scale, bias pre-multiplied by log2 e)
                    psum += p;
                    const _Float16 hi = (_Float16)p;
                    ph[r >> 3][r & 7] = hi;
                    pl[r >> 3][r & 7] = (_Float16)(p - (float)hi);
                }
            }
            l_run = l_run * alpha + psum;
            m_run = m_new;
            // rescale the output accumulators only when some row's running maximum moved (alpha == 1 otherwise: skipping is exact); after the
            // first few key tiles that is rare, and it removes 64 multiplies per tile from a VALU-bound loop
            if (__builtin_amdgcn_ballot_w64(alpha != 1.f) != 0) {
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) { oM[t][r] *= alpha; oC[t][r] *= alpha; }
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const auto pk = __builtin_amdgcn_cvt_pkrtz(sv[r], sv[r + 1]);
                ph[r >> 3][r & 7] = pk[0]; ph[r >> 3][(r & 7) + 1] = pk[1];
                pl[r >> 3][r & 7] = pk[1]; pl[r >> 3][(r & 7) + 1] = pk[0];
            }
            l_run += mx;
        }

        // ---- O^T += V^T P^T.  k-step s covers keys [16s, 16s+16); lane half h owns keys 16s + {0..3} + 4h and 16s + 8 + {0..3} + 4h
        if (VAR & AV_PRIO) __builtin_amdgcn_s_setprio(1);
        if (!(VAR & AV_NOPV)) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                half8 avh[2], avl[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const _Float16* vh = &Vh[cur][(32 * t + qi) * SVLD + 4 * h + 16 * s];
                    const _Float16* vl = &Vl[cur][(32 * t + qi) * SVLD + 4 * h + 16 * s];
                    const half4 h0 = *reinterpret_cast<const half4*>(vh), h1 = *reinterpret_cast<const half4*>(vh + 8);
                    const half4 l0 = *reinterpret_cast<const half4*>(vl), l1 = *reinterpret_cast<const half4*>(vl + 8);
                    avh[t] = half8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
                    avl[t] = half8{l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
                }
                // interleave the two head-dim halves so that consecutive MFMAs never share an accumulator
                oM[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(avh[0], ph[s], oM[0], 0, 0, 0);
                oM[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(avh[1], ph[s], oM[1], 0, 0, 0);
                oC[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(avl[0], ph[s], oC[0], 0, 0, 0);   // V_lo (scaled 2^11) P_hi
                oC[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(avl[1], ph[s], oC[1], 0, 0, 0);
                oM[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(avh[0], pl[s], oM[0], 0, 0, 0);   // V_hi P_lo (unscaled)
                oM[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(avh[1], pl[s], oM[1], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int e = 0; e < 8; ++e) { oM[s][e] += (float)ph[s][e]; oC[s][e] += (float)pl[s][e]; }
        }
        if (VAR & AV_PRIO) __builtin_amdgcn_s_setprio(0);

        __syncthreads();
        cur ^= 1;
    }

    const float l_tot = l_run + xor32(l_run);
    const float inv = 1.f / l_tot;
    if (qvalid) {
        const long orow = (long)b * a.o_bstride + (long)qrow * a.o_qstride + (long)head * a.o_hstride;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d = 32 * t + 8 * g + 4 * h;
                float o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (oM[t][4 * g + j] + oC[t][4 * g + j] * kLoI) * inv;
                if (a.Op) store_planes4(a.Op + ((long)b * a.Nq + qrow) * 2 * (a.H * 64), head * 64 + d, make_float4(o[0], o[1], o[2], o[3]));
                else *reinterpret_cast<float4*>(a.O + orow + d) = make_float4(o[0], o[1], o[2], o[3]);
            }
    }
}

__device__ float g_zero_row[1024];   // stands in for an absent bias (zero-initialised device global; one copy per device)
static const float* zero_row64() {
    static const float* ptr[64] = {};
    int dev = 0;
    HIP_CHECK(hipGetDevice(&dev));
    if (!ptr[dev]) HIP_CHECK(hipGetSymbolAddress((void**)&ptr[dev], HIP_SYMBOL(g_zero_row)));
    return ptr[dev];
}

// ------------------------------------------------------------------------------------------------ ping-pong structure
// One 8-wave workgroup = 256 query rows; waves w and w + 4 share a SIMD.  Every key tile is processed in two barrier-separated phases per wave:
//   M-phase(t): O^T += V^T(t-1) P^T(t-1)  and  S^T(t) = K(t) Q^T      24 MFMAs back to back, nothing else but their LDS fragment reads
//   V-phase(t): online softmax of tile t -> P(t) (hi/lo), staging of the next key tile (global -> registers -> LDS), next bias segment
// and waves 4-7 run ONE PHASE BEHIND waves 0-3 (an extra barrier at their start, one at the others' end): while one wave of a SIMD owns the matrix
// pipe its partner does the VALU work, deterministically, instead of leaving the overlap to wave-age arbitration between two independent
// workgroups (measured on the 4-wave kernel: time = MFMA time + VALU time, i.e. no overlap at all; MI355X guide "two waves per SIMD").
// Waves 0-3 stage the K planes of a tile, waves 4-7 the V^T planes (one 16-byte chunk of each plane per thread): K/V pass through L1/LDS once per
// 256 query rows instead of once per 128.  Two LDS stages: K(t+1) is written in V-phase(t) of waves 0-3 (global phase 2t+1; the slot's previous
// tenant K(t-1) was last read in phase 2t-1), V(t+1) in V-phase(t) of waves 4-7 (phase 2t+2; V(t-1) last read in phase 2t+1, V(t+1) first in 2t+4).
constexpr int PP_PLANE = SKT * SKLD;   // = 64 * SVLD = 2304 halves: one plane of one stage
static_assert(SKT * SKLD == 64 * SVLD, "K and V^T planes share one LDS stage layout");

template <int VAR>
__global__ __launch_bounds__(512) void attention_split_pp_kernel(AttnSplitArgs a) {
    __shared__ __attribute__((aligned(16))) _Float16 lds[2][4][PP_PLANE];   // [stage][Kh, Kl, Vh, Vl]
#ifdef BEVGEN_ATTN_LAB
    __shared__ unsigned long long trace_lds[(VAR & AV_TRACE) ? 8 * 16 * 8 : 1];
#endif

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, grp = wave >> 2;
    const int qi = lane & 31, h = lane >> 5;
    const int head = blockIdx.y, b = blockIdx.z;
    const int qrow = blockIdx.x * 256 + wave * 32 + qi;
    const bool qvalid = qrow < a.Nq;
    const int qc = qvalid ? qrow : a.Nq - 1;

    const long qoff = ((long)b * a.H + head) * a.Nq * 64 + (long)qc * 64 + 8 * h;
    const long koff = ((long)b * a.H + head) * (long)a.Nk_pad * 64;
    const float* Bp = (VAR & AV_BIASPK) ? a.bias_pk + (long)head * a.bias_head_stride + (long)(blockIdx.x * 8 + wave) * a.bias_pk_qb_stride + lane * 4
                                        : a.bias + (long)head * a.bias_head_stride + (long)qc * a.ldbias + 4 * h;
    const int bstep = (VAR & AV_BIASPK) ? a.bias_pk_tile_step : a.bias_tile_step;

    half8 qh[4], ql[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        qh[s] = *reinterpret_cast<const half8*>(a.Qh + qoff + 16 * s);
        ql[s] = *reinterpret_cast<const half8*>(a.Ql + qoff + 16 * s);
    }
    f32x16 oM[2], oC[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) { oM[t][r] = 0.f; oC[t][r] = 0.f; }
    float m_run = kNegBig, l_run = 0.f;

    // staging: this thread's 16-byte chunk of the hi and of the lo plane (K for waves 0-3, V^T for waves 4-7)
    const int st = tid & 255;
    const _Float16 *src_hi, *src_lo;
    long src_step;
    int dst_off;
    if (grp == 0) {
        const int kr = st >> 3, kc = st & 7;
        src_hi = a.Kh + koff + kr * 64 + kc * 8; src_lo = a.Kl + koff + kr * 64 + kc * 8;
        src_step = SKT * 64;
        dst_off = kr * SKLD + kc * 8;
    } else {
        const int vr = st >> 2, vc = st & 3;
        const long o = (VAR & AV_VTILE) ? koff + vr * SKT + vc * 8 : koff + (long)vr * a.Nk_pad + vc * 8;
        src_hi = a.VTh + o; src_lo = a.VTl + o;
        src_step = (VAR & AV_VTILE) ? 64 * SKT : SKT;
        dst_off = 2 * PP_PLANE + vr * SVLD + vc * 8;
    }
    uint4 rh, rl;
    auto gload = [&](int tile) {
        rh = *reinterpret_cast<const uint4*>(src_hi + tile * src_step);
        rl = *reinterpret_cast<const uint4*>(src_lo + tile * src_step);
    };
    auto lstore = [&](int stage) {   // rows are 144 / 72 bytes: 8-byte alignment is what both planes have
        uint2* dh = reinterpret_cast<uint2*>(&lds[stage][0][dst_off]);
        uint2* dl = reinterpret_cast<uint2*>(&lds[stage][1][dst_off]);
        dh[0] = make_uint2(rh.x, rh.y); dh[1] = make_uint2(rh.z, rh.w);
        dl[0] = make_uint2(rl.x, rl.y); dl[1] = make_uint2(rl.z, rl.w);
    };
    f32x16 bacc;
    auto gload_bias = [&](int tile) {
        const float* src = Bp + tile * bstep;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 v = *reinterpret_cast<const float4*>(src + ((VAR & AV_BIASPK) ? 256 * g : 8 * g));
            bacc[4 * g] = v.x; bacc[4 * g + 1] = v.y; bacc[4 * g + 2] = v.z; bacc[4 * g + 3] = v.w;
        }
    };

    const int ntiles = a.Nk_pad / SKT;
    // AV_ONEACC: every product of a k-step accumulates into ONE register set: q_hi 2^-11 and q_lo 2^-11 (= the true low part) are formed once per
    // wave, so that  S = sum k_hi q_hi + k_hi (q_lo 2^-11) + k_lo (q_hi 2^-11)  needs no second accumulator and no 16 fma per tile to merge it
    half8 qs[4], qls[4];
    if (VAR & AV_ONEACC) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int e = 0; e < 8; ++e) { qs[s][e] = qh[s][e] * (_Float16)kLoI; qls[s][e] = ql[s][e] * (_Float16)kLoI; }
    }
    gload_bias(0);
    gload(0);
    lstore(0);
    gload(min(1, ntiles - 1));
    __syncthreads();
    if (grp == 1 && !(VAR & AV_LOCKSTEP)) __syncthreads();   // waves 4-7 start one phase late

    f32x16 sM, sC;
    half8 ph[2], pl[2];
    half8 pvh[2], pvl[2];   // V^T fragments of the first PV k-step, fetched before the barrier in front of the M-phase
    float plate[8];          // AV_SPLITM: p of the tile's second 16 keys, split into hi/lo inside the M-phase (behind the MFMAs of the first k-step)
    float psum_early = 0.f, alpha_keep = 1.f;
    auto vfrag = [&](int stage, int s, half8 (&avh)[2], half8 (&avl)[2]) {
        const _Float16* Vh = &lds[stage][2][0];
        const _Float16* Vl = &lds[stage][3][0];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const _Float16* vh = Vh + (32 * u + qi) * SVLD + 4 * h + 16 * s;
            const _Float16* vl = Vl + (32 * u + qi) * SVLD + 4 * h + 16 * s;
            const half4 h0 = *reinterpret_cast<const half4*>(vh), h1 = *reinterpret_cast<const half4*>(vh + 8);
            const half4 l0 = *reinterpret_cast<const half4*>(vl), l1 = *reinterpret_cast<const half4*>(vl + 8);
            avh[u] = half8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
            avl[u] = half8{l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
        }
    };
    auto pv_step = [&](int s, const half8 (&avh)[2], const half8 (&avl)[2]) {
        oM[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(avh[0], ph[s], oM[0], 0, 0, 0);
        oM[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(avh[1], ph[s], oM[1], 0, 0, 0);
        oC[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(avl[0], ph[s], oC[0], 0, 0, 0);   // V_lo (scaled 2^11) P_hi
        oC[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(avl[1], ph[s], oC[1], 0, 0, 0);
        oM[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(avh[0], pl[s], oM[0], 0, 0, 0);   // V_hi P_lo (unscaled)
        oM[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(avh[1], pl[s], oM[1], 0, 0, 0);
    };
    auto split_pair = [&](float p0, float p1, uint32_t& hwv, uint32_t& lwv) {
        const half2v hp = {(_Float16)p0, (_Float16)p1};
        hwv = __builtin_bit_cast(uint32_t, hp);
        if (VAR & AV_MIX) {
            uint32_t l;
            asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(l) : "v"(p0), "v"(hwv));
            asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(p1), "v"(hwv));
            lwv = l;
        } else {
            const half2v lp = {(_Float16)(p0 - (float)hp[0]), (_Float16)(p1 - (float)hp[1])};
            lwv = __builtin_bit_cast(uint32_t, lp);
        }
    };
    for (int t = 0; t <= ntiles; ++t) {
        // ================================================ M-phase
        PP_STAMP(0);
        if (VAR & AV_PRIO) __builtin_amdgcn_s_setprio(1);
        if (t > 0) {   // O^T += V^T P^T of tile t-1.  k-step s covers keys [16s, 16s+16); lane half h owns keys 16s + {0..3} + 4h and 16s + 8 + {0..3} + 4h
            half8 avh[2], avl[2];
            if (VAR & AV_NOPRELOAD) vfrag((t - 1) & 1, 0, pvh, pvl);
            vfrag((t - 1) & 1, 1, avh, avl);
            pv_step(0, pvh, pvl);
            if (VAR & AV_SPLITM) {   // the second half of P(t-1) and the row sum, in the shadow of the six MFMAs just issued
                uint32_t hw[4], lw[4];
                float ps = psum_early;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    split_pair(plate[2 * r], plate[2 * r + 1], hw[r], lw[r]);
                    ps += plate[2 * r];
                    ps += plate[2 * r + 1];
                }
                ph[1] = __builtin_bit_cast(half8, u32x4{hw[0], hw[1], hw[2], hw[3]});
                pl[1] = __builtin_bit_cast(half8, u32x4{lw[0], lw[1], lw[2], lw[3]});
                l_run = l_run * alpha_keep + ps;
            }
            pv_step(1, avh, avl);
        }
        if (t < ntiles) {   // S^T = K Q^T (+ bias as the C operand of the first main MFMA)
            const _Float16* kh = &lds[t & 1][0][qi * SKLD + 8 * h];
            const _Float16* kl = &lds[t & 1][1][qi * SKLD + 8 * h];
            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (VAR & AV_ONEACC) {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const half8 ah = *reinterpret_cast<const half8*>(kh + 16 * s);
                    const half8 al = *reinterpret_cast<const half8*>(kl + 16 * s);
                    if (s == 0) sM = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, qh[0], bacc, 0, 0, 0);
                    else sM = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, qh[s], sM, 0, 0, 0);
                    sM = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, qls[s], sM, 0, 0, 0);
                    sM = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, qs[s], sM, 0, 0, 0);
                }
            } else {
                {
                    const half8 ah = *reinterpret_cast<const half8*>(kh);
                    const half8 al = *reinterpret_cast<const half8*>(kl);
                    sC = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, ql[0], zero, 0, 0, 0);
                    sM = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, qh[0], bacc, 0, 0, 0);
                    sC = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, qh[0], sC, 0, 0, 0);
                }
#pragma unroll
                for (int s = 1; s < 4; ++s) {
                    const half8 ah = *reinterpret_cast<const half8*>(kh + 16 * s);
                    const half8 al = *reinterpret_cast<const half8*>(kl + 16 * s);
                    sC = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, ql[s], sC, 0, 0, 0);
                    sM = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, qh[s], sM, 0, 0, 0);
                    sC = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, qh[s], sC, 0, 0, 0);
                }
            }
        }
        if (VAR & AV_PRIO) __builtin_amdgcn_s_setprio(0);
        PP_STAMP(1);
        __syncthreads();

        // ================================================ V-phase
        PP_STAMP(2);
        if (VAR & AV_VPRIO) __builtin_amdgcn_s_setprio(2);
        // staging is unconditional (tile indices clamped; a redundant store lands in a stage nobody reads any more): the compiler can then count the
        // loads in flight.  The six loads are spread over the softmax: four waves issuing them back to back fill the CU's one address pipe and the
        // issue of the rest waits behind it (measured: 450 cycles for this block when it sat in one place)
        lstore((t + 1) & 1);
        if (VAR & AV_BUNCH) {
            gload_bias(min(t + 1, ntiles - 1));
            gload(min(t + 2, ntiles - 1));
        }
        PP_STAMP(3);
        if (t < ntiles) {
            float sv[16];
            float mx = kNegBig;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                sv[r] = (VAR & AV_ONEACC) ? sM[r] : fmaf(sC[r], kLoI, sM[r]);
                mx = fmaxf(mx, sv[r]);
            }
            if (!(VAR & AV_BUNCH)) {
                gload_bias(min(t + 1, ntiles - 1));   // (bacc was consumed by the first MFMA of this tile)
                __builtin_amdgcn_sched_barrier(0);
            }
            mx = fmaxf(mx, xor32(mx));
            const float m_new = fmaxf(m_run, mx);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            float psum = 0.f;
            uint32_t hw[8], lw[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const float p0 = __builtin_amdgcn_exp2f(sv[2 * r] - m_new), p1 = __builtin_amdgcn_exp2f(sv[2 * r + 1] - m_new);
                if ((VAR & AV_SPLITM) && r >= 4) {
                    plate[2 * (r - 4)] = p0; plate[2 * (r - 4) + 1] = p1;
                } else {
                    psum += p0;
                    psum += p1;
                    split_pair(p0, p1, hw[r], lw[r]);
                }
                if (r == 3 && !(VAR & AV_BUNCH)) {
                    __builtin_amdgcn_sched_barrier(0);
                    gload(min(t + 2, ntiles - 1));
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            ph[0] = __builtin_bit_cast(half8, u32x4{hw[0], hw[1], hw[2], hw[3]});
            pl[0] = __builtin_bit_cast(half8, u32x4{lw[0], lw[1], lw[2], lw[3]});
            if (VAR & AV_SPLITM) {
                psum_early = psum; alpha_keep = alpha;
            } else {
                ph[1] = __builtin_bit_cast(half8, u32x4{hw[4], hw[5], hw[6], hw[7]});
                pl[1] = __builtin_bit_cast(half8, u32x4{lw[4], lw[5], lw[6], lw[7]});
                l_run = l_run * alpha + psum;
            }
            m_run = m_new;
            if (__builtin_amdgcn_ballot_w64(alpha != 1.f) != 0) {
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int r = 0; r < 16; ++r) { oM[u][r] *= alpha; oC[u][r] *= alpha; }
            }
            if (!(VAR & AV_NOPRELOAD)) vfrag(t & 1, 0, pvh, pvl);   // V(t) has been in LDS since the phase before this one
        }
        if (VAR & AV_VPRIO) __builtin_amdgcn_s_setprio(0);
        PP_STAMP(4);
        __syncthreads();
        PP_STAMP(5);
    }
    if (grp == 0 && !(VAR & AV_LOCKSTEP)) __syncthreads();
#ifdef BEVGEN_ATTN_LAB
    if ((VAR & AV_TRACE) && blockIdx.x == 1 && blockIdx.y == 3 && blockIdx.z == 5)
        for (int i = threadIdx.x; i < 8 * 16 * 8; i += 512) g_attn_trace[i] = trace_lds[i];
#endif

    const float l_tot = l_run + xor32(l_run);
    const float inv = 1.f / l_tot;
    if (qvalid) {
        const long orow = (long)b * a.o_bstride + (long)qrow * a.o_qstride + (long)head * a.o_hstride;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d = 32 * t + 8 * g + 4 * h;
                float o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (oM[t][4 * g + j] + oC[t][4 * g + j] * kLoI) * inv;
                if (a.Op) store_planes4(a.Op + ((long)b * a.Nq + qrow) * 2 * (a.H * 64), head * 64 + d, make_float4(o[0], o[1], o[2], o[3]));
                else *reinterpret_cast<float4*>(a.O + orow + d) = make_float4(o[0], o[1], o[2], o[3]);
            }
    }
}

// bias [Nq, ld] -> packed image [ceil(Nq/32)][Nk_pad/32][4][64][4] (see AV_BIASPK): element (qb, tile, g, lane = qi + 32 h, e) = bias[32 qb + qi][32 tile + 8 g + 4 h + e]
__global__ __launch_bounds__(256) void pack_attn_bias_kernel(const float* __restrict__ bias, int ld, int Nq, int ntiles, float* __restrict__ out, long total4) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;   // one float4 of the image
    if (i >= total4) return;
    const int lane = (int)(i & 63), g = (int)((i >> 6) & 3);
    const long t = i >> 8;
    const int tile = (int)(t % ntiles), qb = (int)(t / ntiles);
    const int q = min(32 * qb + (lane & 31), Nq - 1);
    reinterpret_cast<float4*>(out)[i] = *reinterpret_cast<const float4*>(bias + (long)q * ld + 32 * tile + 8 * g + 4 * (lane >> 5));
}
long attn_bias_packed_floats(int Nq, int Nk_pad) { return (long)cdiv(Nq, 128) * 4 * (Nk_pad / SKT) * 1024; }
void launch_pack_attn_bias(const float* bias, int ld, int Nq, int Nk_pad, float* out, hipStream_t s) {
    BG_REQUIRE(Nk_pad % SKT == 0 && ld % 4 == 0 && ld >= Nk_pad, "pack_attn_bias: Nk_pad=%d ld=%d", Nk_pad, ld);
    const long total4 = attn_bias_packed_floats(Nq, Nk_pad) / 4;
    hipLaunchKernelGGL(pack_attn_bias_kernel, dim3((unsigned)cdiv(total4, 256L)), dim3(256), 0, s, bias, ld, Nq, Nk_pad / SKT, out, total4);
    LAUNCH_CHECK();
}

constexpr int kAttnVariant = 0;
#ifdef BEVGEN_ATTN_LAB
int g_attn_variant = 0, g_attn_extra_lds = 0;
void attn_lab_read_trace(unsigned long long* out) { HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_attn_trace), sizeof(g_attn_trace))); }
#endif

template <int VAR>
static void launch_variant(const AttnSplitArgs& a, dim3 grid, int extra_lds, hipStream_t s) {
    hipLaunchKernelGGL(attention_split_kernel<VAR>, grid, dim3(256), extra_lds, s, a);
}
template <int VAR>
static void launch_variant_pp(const AttnSplitArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(attention_split_pp_kernel<VAR>, dim3(cdiv(a.Nq, 256), a.H, a.B), dim3(512), 0, s, a);
}

void launch_attention_split(const AttnSplitArgs& a0, hipStream_t s) {
    AttnSplitArgs a = a0;
    BG_REQUIRE(a.Nk_pad % SKT == 0 && a.Nk_pad > 0, "attention_split: Nk_pad=%d must be a positive multiple of %d", a.Nk_pad, SKT);
    BG_REQUIRE(a.bias == nullptr || a.ldbias % 4 == 0, "attention_split: bias row stride must be a multiple of 4");
    a.bias_tile_step = SKT;
    a.bias_pk_tile_step = 1024;
    a.bias_pk_qb_stride = (long)(a.Nk_pad / SKT) * 1024;
    if (a.bias_pk == nullptr) a.bias_pk = a.bias;   // (lab: unpacked variants never read it)
    if (a.bias == nullptr) {   // unconditional bias loads: a zero block, no strides
        a.bias = a.bias_pk = zero_row64();
        a.ldbias = 0; a.bias_head_stride = 0; a.bias_tile_step = 0; a.bias_pk_tile_step = 0; a.bias_pk_qb_stride = 0;
    }
    dim3 grid(cdiv(a.Nq, 128), a.H, a.B);
    ProfScope prof(PROF_ATTN, 4.0 * a.B * a.H * (double)a.Nq * a.Nk_pad * 64, s);
#ifdef BEVGEN_ATTN_LAB
    switch (g_attn_variant) {
#define AV_CASE(v) case v: launch_variant<v>(a, grid, g_attn_extra_lds, s); break;
        AV_CASE(0) AV_CASE(1) AV_CASE(2) AV_CASE(3) AV_CASE(8) AV_CASE(9) AV_CASE(16) AV_CASE(24) AV_CASE(25) AV_CASE(27) AV_CASE(31) AV_CASE(192) AV_CASE(216)
#undef AV_CASE
#define PP_CASE(v) case 1024 + v: launch_variant_pp<v>(a, s); break;
        PP_CASE(4096 + 256 + 10) PP_CASE(65536 + 4096 + 256 + 10) PP_CASE(4096 + 256 + 10 + AV_TRACE) PP_CASE(65536 + 4096 + 256 + 10 + AV_TRACE)
#undef PP_CASE
        default: BG_REQUIRE(false, "attention lab: variant %d not instantiated", g_attn_variant);
    }
#else
    launch_variant<kAttnVariant>(a, grid, 0, s);
#endif
    LAUNCH_CHECK();
}

// ------------------------------------------------------------------------------------------------ q / kv preparation with split outputs
// (muse_maskgit_pytorch.py:132-146: x8, null-kv concat, l2norm eps 1e-12, q_scale / k_scale), then hi/lo split; V is written transposed.
__device__ __forceinline__ void split1(float v, _Float16& hi, _Float16& lo) {
    hi = (_Float16)v;
    lo = (_Float16)((v - (float)hi) * kLo);
}

__global__ __launch_bounds__(256) void muse_q_prep_split_kernel(const float* __restrict__ qraw, const float* __restrict__ q_scale, _Float16* __restrict__ Qh,
                                                                _Float16* __restrict__ Ql, int H, int Nq, long total, float post) {
    const int lane = threadIdx.x & 63;
    const long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= total) return;
    const int h = (int)(w % H);
    const long bn = w / H;
    const int n = (int)(bn % Nq);
    const long b = bn / Nq;
    const float v = qraw[bn * (H * 64) + h * 64 + lane] * 8.0f;
    const float nrm = fmaxf(sqrtf(wave_sum(v * v)), 1e-12f);
    const float q = ((v / nrm) * q_scale[lane]) * post;
    _Float16 hi, lo;
    split1(q, hi, lo);
    const long dst = ((b * H + h) * Nq + n) * 64 + lane;
    Qh[dst] = hi;
    Ql[dst] = lo;
}

void launch_muse_q_prep_split(const float* qraw, const float* q_scale, void* Qh, void* Ql, int B, int H, int Nq, float post, hipStream_t s) {
    const long total = (long)B * Nq * H;
    hipLaunchKernelGGL(muse_q_prep_split_kernel, dim3((int)((total + 3) / 4)), dim3(256), 0, s, qraw, q_scale, reinterpret_cast<_Float16*>(Qh),
                       reinterpret_cast<_Float16*>(Ql), H, Nq, total, post);
    LAUNCH_CHECK();
}

// one workgroup = 64 consecutive key rows of one (batch, head): K rows are written row-major, V goes through LDS so that the transposed
// [64 dims][keys] image is written with contiguous 128-byte segments
__global__ __launch_bounds__(256) void muse_kv_prep_split_kernel(const float* __restrict__ kvraw, const float* __restrict__ null_kv, const float* __restrict__ k_scale,
                                                                 _Float16* __restrict__ Kh, _Float16* __restrict__ Kl, _Float16* __restrict__ VTh,
                                                                 _Float16* __restrict__ VTl, int H, int Nk, int Nk_pad) {
    __shared__ float vt[64][65];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int j0 = blockIdx.x * 64, h = blockIdx.y;
    const long b = blockIdx.z;
    const long kbase = (b * H + h) * (long)Nk_pad * 64;
    for (int jj = wv; jj < 64; jj += 4) {
        const int j = j0 + jj;  // key row incl. the null key at j = 0
        float kv = 0.f, vv = 0.f;
        if (j <= Nk) {
            if (j == 0) {
                kv = null_kv[h * 64 + lane];
                vv = null_kv[(long)H * 64 + h * 64 + lane];
            } else {
                const float* src = kvraw + (b * Nk + (j - 1)) * (2L * H * 64);
                kv = src[h * 64 + lane];
                vv = src[H * 64 + h * 64 + lane];
            }
            const float nrm = fmaxf(sqrtf(wave_sum(kv * kv)), 1e-12f);
            const float k = (kv / nrm) * k_scale[lane];
            _Float16 hi, lo;
            split1(k, hi, lo);
            Kh[kbase + (long)j * 64 + lane] = hi;
            Kl[kbase + (long)j * 64 + lane] = lo;
        }
        vt[jj][lane] = vv;  // rows beyond Nk stay zero
    }
    __syncthreads();
    for (int d = wv; d < 64; d += 4) {
        const int j = j0 + lane;
        if (j < Nk_pad) {
            _Float16 hi, lo;
            split1(vt[lane][d], hi, lo);
            VTh[kbase + (long)d * Nk_pad + j] = hi;
            VTl[kbase + (long)d * Nk_pad + j] = lo;
        }
    }
}

// prepared null key / value of one attention module for the fused to_kv epilogue (EPI_MUSE_KV): out = [k_hi | k_lo | v_hi | v_lo][H][64] halves
__global__ __launch_bounds__(64) void muse_null_kv_prep_kernel(const float* __restrict__ null_kv, const float* __restrict__ k_scale, _Float16* __restrict__ out, int H) {
    const int h = blockIdx.x, lane = threadIdx.x;
    const float kv = null_kv[h * 64 + lane], vv = null_kv[(long)H * 64 + h * 64 + lane];
    const float nrm = fmaxf(sqrtf(wave_sum(kv * kv)), 1e-12f);
    const float k = (kv / nrm) * k_scale[lane];
    _Float16 hi, lo;
    split1(k, hi, lo);
    out[h * 64 + lane] = hi;
    out[H * 64 + h * 64 + lane] = lo;
    split1(vv, hi, lo);
    out[2 * H * 64 + h * 64 + lane] = hi;
    out[3 * H * 64 + h * 64 + lane] = lo;
}
void launch_muse_null_kv_prep(const float* null_kv, const float* k_scale, void* out, int H, hipStream_t s) {
    hipLaunchKernelGGL(muse_null_kv_prep_kernel, dim3(H), dim3(64), 0, s, null_kv, k_scale, reinterpret_cast<_Float16*>(out), H);
    LAUNCH_CHECK();
}

void launch_muse_kv_prep_split(const float* kvraw, const float* null_kv, const float* k_scale, void* Kh, void* Kl, void* VTh, void* VTl, int B, int H, int Nk,
                               int Nk_pad, hipStream_t s) {
    dim3 grid(cdiv(Nk_pad, 64), H, B);
    hipLaunchKernelGGL(muse_kv_prep_split_kernel, grid, dim3(256), 0, s, kvraw, null_kv, k_scale, reinterpret_cast<_Float16*>(Kh), reinterpret_cast<_Float16*>(Kl),
                       reinterpret_cast<_Float16*>(VTh), reinterpret_cast<_Float16*>(VTl), H, Nk, Nk_pad);
    LAUNCH_CHECK();
}

}  // namespace bevgen
