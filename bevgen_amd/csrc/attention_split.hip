// Split-precision flash attention (head dim 64) for Route M:  O = softmax(scale * Q K^T + bias) V  (scale and bias arrive multiplied by log2 e and
// the exponentials are v_exp_f32 = 2^x) with fp32-class accuracy on the f16 matrix cores.  Scores are computed transposed so the query index sits on
// the lane axis: the softmax is in-lane plus one xor-32 exchange, and exp(S^T) is already the B operand of O^T = V^T P^T.  Every matrix product is
//        X Y^T ~= hi_x hi_y^T + 2^-11 (hi_x lo_y^T + lo_x hi_y^T),      x = hi + lo * 2^-11 (two f16 numbers, 22 mantissa bits)
// with THREE v_mfma_f32_32x32x16_f16 per 16-deep k-step instead of EIGHT v_mfma_f32_32x32x2_f32 (see gemm_split.hip).
//   * Q, K arrive pre-split from the q / k preparation (fused GEMM epilogues), V pre-split AND transposed ([B,H,64,Nk_pad]) so that the V^T operand
//     rows are contiguous along the key axis;
//   * the three products of a QK^T k-step accumulate into ONE register set: q_hi 2^-11 and q_lo 2^-11 (the true low part; f16 subnormals are honoured
//     by the MFMA) are formed once per wave, and the bias segment is the C operand of the first MFMA (its register layout IS the accumulator layout):
//     no second accumulator, no merge, no bias add in the softmax;
//   * P = exp(S - m) is split in registers: one v_cvt_pk_f16_f32 per pair for the high parts, one v_fma_mix{lo,hi}_f16 per element for
//     lo = p - hi (computed in fp32, rounded to f16, written into the packed word in place);
//   * the score registers of lane half h hold keys {(r&3) + 8(r>>2) + 4h}, so for the k-step that covers keys [16s, 16s+16) half h owns keys
//     16s + {0..3, 8..11} + 4h - the V^T fragment is gathered with the same permutation (two 8-byte LDS reads), which is legal because any k
//     permutation applied to both MFMA operands leaves the product unchanged;
//   * LDS rows are padded (K: 72 halves, V^T: 36 halves) so that ds_read_b128 / ds_read_b64 lane groups are bank-conflict free;
//   * the bias is read from a packed image [q block of 32][key tile][4][64 lanes][4] (launch_pack_attn_bias, built once per context): every load
//     instruction of a wave is 1 KiB contiguous (the row-major matrix gave 32 pieces of 32 bytes per instruction).
//
// PING-PONG STRUCTURE.  One 8-wave workgroup = 256 query rows; waves w and w + 4 share a SIMD.  Every key tile is two barrier-separated phases:
//   M-phase(t): O^T += V^T(t-1) P^T(t-1)  and  S^T(t) = K(t) Q^T      24 MFMAs, nothing else but their LDS fragment reads
//   V-phase(t): online softmax of tile t -> P(t) (hi/lo), staging of the next key tile (global -> registers -> LDS), next bias segment
// and waves 4-7 run ONE PHASE BEHIND waves 0-3 (an extra barrier at their start, one at the others' end): while one wave of a SIMD owns the matrix
// pipe its partner does the VALU work, deterministically.  The round-1 kernel (4 waves, two independent workgroups per CU, kept below for
// tools/attn_lab) left that overlap to wave-age arbitration and got none: its time was MFMA time + everything else (removing all MFMAs: 651 -> 466 us,
// the MFMAs alone are 215 us; profiles/r02_attention_lab.txt).  Waves 0-3 stage the K planes of a tile, waves 4-7 the V^T planes (one 16-byte chunk
// of each plane per thread): K/V pass through L1 / LDS once per 256 query rows.  Two LDS stages: K(t+1) is written in V-phase(t) of waves 0-3
// (global phase 2t+1; the slot's previous tenant K(t-1) was last read in phase 2t-1), V(t+1) in V-phase(t) of waves 4-7 (phase 2t+2; V(t-1) last
// read in phase 2t+1, V(t+1) first in 2t+4).  Measured on BASELINE configs[1]'s self-attention shape (16 scenes x 16 heads, 1536 x 1568):
// 651 -> 528 us.  Tried on top and rejected by measurement (same file): s_setprio on either phase, tile-major V^T, packed-f32 VALU, fetching the
// first V^T fragments before the barrier, splitting half of P inside the M-phase, both waves of a SIMD in the same phase (597 us).
#include <algorithm>

#include "common.h"
#include "kernels.h"
#include "profiler.h"

namespace bevgen {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int SKT = 32;    // keys per tile
constexpr int SKLD = 72;   // K tile row stride (halves)
constexpr int SVLD = 36;   // V^T tile row stride (halves)
constexpr float kLo = 2048.f, kLoI = 1.f / 2048.f;
constexpr int PP_PLANE = SKT * SKLD;   // = 64 * SVLD = 2304 halves: one plane of one LDS stage
static_assert(SKT * SKLD == 64 * SVLD, "K and V^T planes share one LDS stage layout");

// tools/attn_lab only (-DBEVGEN_ATTN_DIAG=bits; wrong results by design, what each part of the loop costs): 1 no softmax arithmetic, 2 no tile / bias loads and no LDS
// stores in the loop, 4 no matrix instructions, 8 no bias loads, 16 no tile loads / LDS stores.  0 (the product build) compiles every test away.
#ifndef BEVGEN_ATTN_DIAG
#define BEVGEN_ATTN_DIAG 0
#endif
// 1: tile and bias loads in the buffer form (wave-uniform resource + a per-lane 32-bit offset that never changes + a scalar tile offset: no 64-bit address arithmetic, the
// four bias requests differ in a scalar only).  Lab, same box: 527 -> 521 us (profiles/r06_attn_lab_diag.txt; a 4-byte placement shift of the loop changes nothing: 520-523 us
// at all four phases).  0 keeps the global_load form - note that THAT build is 12 % slower than the same form before this switch existed (586 us): two register copies and
// one s_waitcnt land in the middle of the softmax and expose the tile loads' latency; the kernel's time hangs on where its waits fall.
#ifndef BEVGEN_ATTN_BUF
#define BEVGEN_ATTN_BUF 1
#endif
#ifdef BEVGEN_ATTN_LAB
__device__ unsigned long long g_attn_trace[8 * 16 * 8];   // [wave][tile][stamp]
#define PP_STAMP(k) do { if (TRACE && lane == 0 && t < 16) trace_lds[(wave * 16 + t) * 8 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define PP_STAMP(k) do { } while (0)
#endif

// KS: key-split form (AttnSplitArgs::ksplit > 1; a template flag so that the throughput instantiation carries none of its index arithmetic)
template <bool TRACE, bool KS = false>
__global__ __launch_bounds__(512) void attention_split_kernel(AttnSplitArgs a) {
    __shared__ __attribute__((aligned(16))) _Float16 lds[2][4][PP_PLANE];   // [stage][Kh, Kl, Vh, Vl]
    __shared__ __attribute__((aligned(16))) float ostage[8][32 * 68];   // output rows of a wave on their way to row-major stores (epilogue)
#ifdef BEVGEN_ATTN_LAB
    __shared__ unsigned long long trace_lds[TRACE ? 8 * 16 * 8 : 1];
#endif

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, grp = wave >> 2;
    const int qi = lane & 31, h = lane >> 5;
    // XCD-aware order: workgroup L runs on XCD L % 8 (each XCD has its own L2).  The q blocks of one (batch, head) read the same K / V planes, so they
    // are given to ONE XCD: virtual id v = (L % 8) * (total / 8) + L / 8, q block = v % nqb - K / V enter one L2 instead of (up to) all eight
    const int nqb = gridDim.x, total = nqb * gridDim.y * gridDim.z;
    const int lin = blockIdx.x + nqb * (blockIdx.y + gridDim.y * blockIdx.z);
    const int vid = (total % 8 == 0) ? (lin % 8) * (total / 8) + lin / 8 : lin;
    const int qblk = vid % nqb, head = (vid / nqb) % gridDim.y, bz = vid / (nqb * gridDim.y);
    const int nks = KS ? a.ksplit : 1;
    const int b = KS ? bz / nks : bz, ks = KS ? bz - b * nks : 0;   // gridDim.z = B * ksplit
    const int qrow = qblk * 256 + wave * 32 + qi;
    const bool qvalid = qrow < a.Nq;
    const int qc = qvalid ? qrow : a.Nq - 1;

    const long qoff = ((long)b * a.H + head) * a.Nq * 64 + (long)qc * 64 + 8 * h;
    const long koff = ((long)(b / a.kv_group) * a.H + head) * (long)a.Nk_pad * 64;
    // (no bias: the launcher points bias_pk at a zero block with both steps 0 - unconditional loads, no select in the loop)
    const int bstep = a.bias_pk_tile_step;
    // this workgroup's key tiles [t_first, t_first + ntiles)
    const int ntiles_all = a.Nk_pad / SKT;
    const int t_first = KS ? ks * ntiles_all / nks : 0;
    const int ntiles = KS ? (ks + 1) * ntiles_all / nks - t_first : ntiles_all;
    const float* Bp = a.bias_pk + (long)head * a.bias_head_stride + (long)(qblk * 8 + wave) * a.bias_pk_qb_stride + lane * 4 + (long)t_first * bstep;

    // q_hi, q_hi 2^-11, q_lo 2^-11
    half8 qh[4], qs[4], qls[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        qh[s] = *reinterpret_cast<const half8*>(a.Qh + qoff + 16 * s);
        const half8 ql = *reinterpret_cast<const half8*>(a.Ql + qoff + 16 * s);
#pragma unroll
        for (int e = 0; e < 8; ++e) { qs[s][e] = qh[s][e] * (_Float16)kLoI; qls[s][e] = ql[e] * (_Float16)kLoI; }
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
    int src_step, dst_off;
    if (grp == 0) {
        const int kr = st >> 3, kc = st & 7;
        src_step = SKT * 64;
        src_hi = a.Kh + koff + kr * 64 + kc * 8 + (long)t_first * src_step; src_lo = a.Kl + koff + kr * 64 + kc * 8 + (long)t_first * src_step;
        dst_off = kr * SKLD + kc * 8;
    } else {
        const int vr = st >> 2, vc = st & 3;
        src_step = SKT;
        src_hi = a.VTh + koff + (long)vr * a.Nk_pad + vc * 8 + (long)t_first * src_step; src_lo = a.VTl + koff + (long)vr * a.Nk_pad + vc * 8 + (long)t_first * src_step;
        dst_off = 2 * PP_PLANE + vr * SVLD + vc * 8;
    }
    uint4 rh, rl;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    // buffer form: K planes for waves 0-3, V^T planes for waves 4-7 (wave-uniform bases); the lane's piece is a constant byte offset, the tile a scalar one
    const int step_u = wave_u < 4 ? SKT * 64 : SKT;   // (= src_step, as a scalar)
    const _Float16* tb_hi = (wave_u < 4 ? a.Kh : a.VTh) + koff + (long)t_first * step_u;
    const _Float16* tb_lo = (wave_u < 4 ? a.Kl : a.VTl) + koff + (long)t_first * step_u;
    const __amdgpu_buffer_rsrc_t rs_hi = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(tb_hi), 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_lo = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(tb_lo), 0, -1, 0x00020000);
    const int tv_off = wave_u < 4 ? ((st >> 3) * 64 + (st & 7) * 8) * 2 : ((st >> 2) * a.Nk_pad + (st & 3) * 8) * 2;
    auto gload = [&](int tile) {
        if (BEVGEN_ATTN_DIAG & (2 | 16)) return;
        if (BEVGEN_ATTN_BUF) {
            const int so = tile * step_u * 2;
            rh = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs_hi, tv_off, so, 0));
            rl = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs_lo, tv_off, so, 0));
            return;
        }
        rh = *reinterpret_cast<const uint4*>(src_hi + (long)tile * src_step);
        rl = *reinterpret_cast<const uint4*>(src_lo + (long)tile * src_step);
    };
    auto lstore = [&](int stage) {   // rows are 144 / 72 bytes: 8-byte alignment is what both planes have
        if (BEVGEN_ATTN_DIAG & (2 | 16)) return;
        uint2* dh = reinterpret_cast<uint2*>(&lds[stage][0][dst_off]);
        uint2* dl = reinterpret_cast<uint2*>(&lds[stage][1][dst_off]);
        dh[0] = make_uint2(rh.x, rh.y); dh[1] = make_uint2(rh.z, rh.w);
        dl[0] = make_uint2(rl.x, rl.y); dl[1] = make_uint2(rl.z, rl.w);
    };
    f32x16 bacc;   // bias segment of the next tile: this lane's 16 keys in accumulator order
    if (BEVGEN_ATTN_DIAG) {   // (defined values for the parts a diagnostic build leaves out)
        for (int r = 0; r < 16; ++r) bacc[r] = (float)(lane + r);
        rh = make_uint4(lane, 1, 2, 3); rl = rh;
    }
    const float* Bu = a.bias_pk + (long)head * a.bias_head_stride + (long)(qblk * 8 + wave_u) * a.bias_pk_qb_stride + (long)t_first * bstep;   // wave-uniform part of Bp
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Bu), 0, -1, 0x00020000);
    auto gload_bias = [&](int tile) {
        if (BEVGEN_ATTN_DIAG & (2 | 8)) return;
        if (BEVGEN_ATTN_BUF) {
            const int so = tile * bstep * 4;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_b, lane * 16, so + 1024 * g, 0);
                bacc[4 * g] = __uint_as_float(v.x); bacc[4 * g + 1] = __uint_as_float(v.y); bacc[4 * g + 2] = __uint_as_float(v.z); bacc[4 * g + 3] = __uint_as_float(v.w);
            }
            return;
        }
        const float* src = Bp + (long)tile * bstep;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 v = *reinterpret_cast<const float4*>(src + 256 * g);
            bacc[4 * g] = v.x; bacc[4 * g + 1] = v.y; bacc[4 * g + 2] = v.z; bacc[4 * g + 3] = v.w;
        }
    };

    gload_bias(0);
    gload(0);
    lstore(0);
    gload(min(1, ntiles - 1));
    __syncthreads();
    if (grp == 1) __syncthreads();   // waves 4-7 start one phase late

    f32x16 sM;
    half8 ph[2], pl[2];
    for (int t = 0; t <= ntiles; ++t) {
        // ================================================ M-phase
        PP_STAMP(0);
        if (t > 0 && !(BEVGEN_ATTN_DIAG & 4)) {   // O^T += V^T P^T of tile t-1.  k-step s covers keys [16s, 16s+16); lane half h owns keys 16s + {0..3} + 4h and 16s + 8 + {0..3} + 4h
            const _Float16* Vh = &lds[(t - 1) & 1][2][0];
            const _Float16* Vl = &lds[(t - 1) & 1][3][0];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                half8 avh[2], avl[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const _Float16* vh = Vh + (32 * u + qi) * SVLD + 4 * h + 16 * s;
                    const _Float16* vl = Vl + (32 * u + qi) * SVLD + 4 * h + 16 * s;
                    const half4 h0 = *reinterpret_cast<const half4*>(vh), h1 = *reinterpret_cast<const half4*>(vh + 8);
                    const half4 l0 = *reinterpret_cast<const half4*>(vl), l1 = *reinterpret_cast<const half4*>(vl + 8);
                    avh[u] = half8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
                    avl[u] = half8{l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
                }
                // interleave the two head-dim halves so that consecutive MFMAs never share an accumulator
                oM[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(avh[0], ph[s], oM[0], 0, 0, 0);
                oM[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(avh[1], ph[s], oM[1], 0, 0, 0);
                oC[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(avl[0], ph[s], oC[0], 0, 0, 0);   // V_lo (scaled 2^11) P_hi
                oC[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(avl[1], ph[s], oC[1], 0, 0, 0);
                oM[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(avh[0], pl[s], oM[0], 0, 0, 0);   // V_hi P_lo (unscaled)
                oM[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(avh[1], pl[s], oM[1], 0, 0, 0);
            }
        }
        if (t < ntiles && !(BEVGEN_ATTN_DIAG & 4)) {   // S^T = bias + K Q^T
            const _Float16* kh = &lds[t & 1][0][qi * SKLD + 8 * h];
            const _Float16* kl = &lds[t & 1][1][qi * SKLD + 8 * h];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const half8 ah = *reinterpret_cast<const half8*>(kh + 16 * s);
                const half8 al = *reinterpret_cast<const half8*>(kl + 16 * s);
                if (s == 0) sM = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, qh[0], bacc, 0, 0, 0);
                else sM = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, qh[s], sM, 0, 0, 0);
                sM = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, qls[s], sM, 0, 0, 0);
                sM = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, qs[s], sM, 0, 0, 0);
            }
        }
        if ((BEVGEN_ATTN_DIAG & 4) && t < ntiles) { for (int r = 0; r < 16; ++r) sM[r] = bacc[r] + (float)t; asm volatile("" : "+v"(sM)); }
        PP_STAMP(1);
        __syncthreads();

        // ================================================ V-phase
        PP_STAMP(2);
        // staging is unconditional (tile indices clamped; a redundant store lands in a stage nobody reads any more): the compiler can then count the
        // loads in flight, and the wait in front of the first QK^T MFMA is for the bias segment only.  The six loads are spread over the softmax:
        // four waves issuing them back to back fill the CU's one address pipe and the rest of the wave's instructions wait behind it
        lstore((t + 1) & 1);
        PP_STAMP(3);
        if (t < ntiles) {
#if BEVGEN_ATTN_DIAG & 1
            gload_bias(min(t + 1, ntiles - 1));
            gload(min(t + 2, ntiles - 1));
            const float m_new = m_run, alpha = 1.f, psum = 1.f;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                ph[s] = __builtin_bit_cast(half8, u32x4{__float_as_uint(sM[8 * s]), __float_as_uint(sM[8 * s + 1]), __float_as_uint(sM[8 * s + 2]), __float_as_uint(sM[8 * s + 3])});
                pl[s] = __builtin_bit_cast(half8, u32x4{__float_as_uint(sM[8 * s + 4]), __float_as_uint(sM[8 * s + 5]), __float_as_uint(sM[8 * s + 6]), __float_as_uint(sM[8 * s + 7])});
            }
#else
            float mx = kNegBig;
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sM[r]);
            gload_bias(min(t + 1, ntiles - 1));   // (bacc was consumed by the first MFMA of this tile)
            __builtin_amdgcn_sched_barrier(0);
            mx = fmaxf(mx, xor32(mx));
            // (Round 4, measured and rejected: a LAZY reference - a row's reference only moves when its tile maximum is more than 2^14 above it, so that the rescale pass
            // below almost never runs.  On tools/attn_lab's random scores, where some row of the wave sees a new maximum on nearly every tile and the rescale took
            // 400-550 of the V-phase's 1450 clocks, 527 -> 483 us; on the bench's model, where maxima settle within the first tiles and the pass is skipped anyway,
            // 278 -> 248 TF-equiv in same-box A/B (profiles/r04_ab_attn_lazy_reference.txt): same instruction stream, but p no longer lies in [0, 1] - larger hi / lo
            // operand magnitudes on a power-limited matrix pipe.  Likewise rejected: the bias segment two tiles ahead by LDS-DMA instead of one tile ahead in
            // registers, 545 -> 596 us - the V-phase is not waiting for that load, and the ring's 8 KiB per wave and tile of extra LDS traffic costs 9 %.)
            const float m_new = fmaxf(m_run, mx);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            PP_STAMP(6);
            float psum = 0.f;
            uint32_t hw[8], lw[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                // scores are in the base-2 domain (scale, bias pre-multiplied by log2 e)
                const float p0 = __builtin_amdgcn_exp2f(sM[2 * r] - m_new), p1 = __builtin_amdgcn_exp2f(sM[2 * r + 1] - m_new);
                psum += p0;
                psum += p1;
                // p in [0,1] = hi + lo with lo stored UNSCALED: |lo| <= 2^-12, and what the f16 subnormal range drops is below 2^-25 ABSOLUTE, which
                // is what matters for a probability (a weight of the row sum); V_hi P_lo then accumulates straight into oM
                const half2v hp = {(_Float16)p0, (_Float16)p1};
                hw[r] = __builtin_bit_cast(uint32_t, hp);
                uint32_t l;
                asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(l) : "v"(p0), "v"(hw[r]));
                asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(p1), "v"(hw[r]));
                lw[r] = l;
                if (r == 3) {
                    __builtin_amdgcn_sched_barrier(0);
                    gload(min(t + 2, ntiles - 1));
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                ph[s] = __builtin_bit_cast(half8, u32x4{hw[4 * s], hw[4 * s + 1], hw[4 * s + 2], hw[4 * s + 3]});
                pl[s] = __builtin_bit_cast(half8, u32x4{lw[4 * s], lw[4 * s + 1], lw[4 * s + 2], lw[4 * s + 3]});
            }
#endif
            PP_STAMP(7);
            l_run = l_run * alpha + psum;
            m_run = m_new;
            // rescale the output accumulators only when some row's running maximum moved (alpha == 1 otherwise: skipping is exact); after the
            // first few key tiles that is rare
            if (__builtin_amdgcn_ballot_w64(alpha != 1.f) != 0) {
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int r = 0; r < 16; ++r) { oM[u][r] *= alpha; oC[u][r] *= alpha; }
            }
        }
        PP_STAMP(4);
        __syncthreads();
        PP_STAMP(5);
    }
    if (grp == 0) __syncthreads();
#ifdef BEVGEN_ATTN_LAB
    if (TRACE && blockIdx.x == 1 && blockIdx.y == 3 && blockIdx.z == 5)
        for (int i = threadIdx.x; i < 8 * 16 * 8; i += 512) g_attn_trace[i] = trace_lds[i];
#endif

    const float l_tot = l_run + xor32(l_run);
    if (KS) {   // partial result of this key range: unnormalised output row (relative to m_run), m_run, row sum - rows of kAttnPartPitch floats, stored row-major like the planes below
        float* st = ostage[wave];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = oM[t][4 * g + j] + oC[t][4 * g + j] * kLoI;
                *reinterpret_cast<f32x4*>(st + qi * 68 + 32 * t + 8 * g + 4 * h) = o;
            }
        if (h == 0) *reinterpret_cast<float2*>(st + qi * 68 + 64) = make_float2(m_run, l_tot);
        const int c = lane & 15, rr = lane >> 4;
#pragma unroll
        for (int ps = 0; ps < 8; ++ps) {
            const int row = ps * 4 + rr, q = qblk * 256 + wave * 32 + row;
            const f32x4 v = *reinterpret_cast<const f32x4*>(st + row * 68 + 4 * c);
            if (q < a.Nq) {
                float* wrow = a.kws + ((((long)ks * a.B + b) * a.H + head) * a.Nq + q) * kAttnPartPitch;
                *reinterpret_cast<f32x4*>(wrow + 4 * c) = v;
                if (c == 0) *reinterpret_cast<float2*>(wrow + 64) = *reinterpret_cast<const float2*>(st + row * 68 + 64);
            }
        }
        return;
    }
    const float inv = 1.f / l_tot;
    if (!KS && a.Op) {
        // Plane output, row-major.  In the accumulator layout a store instruction is 64 rows x 8 bytes - 64 partial lines - and sixteen of them per wave held the CU
        // ~6 us per workgroup with nothing else to run (the LDS-DMA GEMM's epilogue had the same disease: tools/storebw, experiments/r06.md).  A (row, head) pair is
        // 256 contiguous bytes of the plane image ([hi 32 | lo 32] of dims 0-31, then of dims 32-63): the wave's 32 rows go through its own LDS slice (row stride 68
        // floats: conflict-free both ways) and leave as 16 lanes x 16 bytes per row, 4 rows per instruction - 8 store instructions of whole lines.  Same arithmetic per
        // element: bit-identical output.
        float* st = ostage[wave];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (oM[t][4 * g + j] + oC[t][4 * g + j] * kLoI) * inv;
                *reinterpret_cast<f32x4*>(st + qi * 68 + 32 * t + 8 * g + 4 * h) = o;
            }
        const int c = lane & 15, rr = lane >> 4;
        const int dsel = (c >> 3) * 32 + (c & 3) * 8;   // this lane's 8 dims; lanes with bit 2 of c set store their lo parts
        const bool lo_sel = (c >> 2) & 1;
        unsigned bad = 0;
#pragma unroll
        for (int ps = 0; ps < 8; ++ps) {
            const int row = ps * 4 + rr, q = qblk * 256 + wave * 32 + row;
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(st + row * 68 + dsel);
            const f32x4 a1 = *reinterpret_cast<const f32x4*>(st + row * 68 + dsel + 4);
            half8 hi8, lo8;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                hi8[e] = split_hi(a0[e]); lo8[e] = split_lo(a0[e], hi8[e]);
                hi8[4 + e] = split_hi(a1[e]); lo8[4 + e] = split_lo(a1[e], hi8[4 + e]);
            }
            if (q < a.Nq) {
                if (!lo_sel) guard_half8(hi8, bad);
                *reinterpret_cast<half8*>(a.Op + ((long)b * a.Nq + q) * 2 * (a.H * 64) + head * 128 + c * 8) = lo_sel ? lo8 : hi8;
            }
        }
        if (bad) status_raise(a.status, BG_ST_F16_RANGE);
        return;
    }
    if (qvalid) {
        unsigned bad = 0;
        const long orow = (long)b * a.o_bstride + (long)qrow * a.o_qstride + (long)head * a.o_hstride;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d = 32 * t + 8 * g + 4 * h;
                float o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (oM[t][4 * g + j] + oC[t][4 * g + j] * kLoI) * inv;
                if (a.Op) store_planes4(a.Op + ((long)b * a.Nq + qrow) * 2 * (a.H * 64), head * 64 + d, make_float4(o[0], o[1], o[2], o[3]), bad);
                else *reinterpret_cast<float4*>(a.O + orow + d) = make_float4(o[0], o[1], o[2], o[3]);
            }
        if (bad) status_raise(a.status, BG_ST_F16_RANGE);
    }
}

// bias [Nq, ld] -> packed image [ceil(Nq/256) * 8 q blocks][Nk_pad/32][4][64][4]: element (qb, tile, g, lane = qi + 32 h, e) = bias[32 qb + qi][32 tile + 8 g + 4 h + e]
__global__ __launch_bounds__(256) void pack_attn_bias_kernel(const float* __restrict__ bias, int ld, int Nq, int ntiles, float* __restrict__ out, long total4) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;   // one float4 of the image
    if (i >= total4) return;
    const int lane = (int)(i & 63), g = (int)((i >> 6) & 3);
    const long t = i >> 8;
    const int tile = (int)(t % ntiles), qb = (int)(t / ntiles);
    const int q = min(32 * qb + (lane & 31), Nq - 1);
    reinterpret_cast<float4*>(out)[i] = *reinterpret_cast<const float4*>(bias + (long)q * ld + 32 * tile + 8 * g + 4 * (lane >> 5));
}
long attn_bias_packed_floats(int Nq, int Nk_pad) { return (long)cdiv(Nq, 256) * 8 * (Nk_pad / SKT) * 1024; }
void launch_pack_attn_bias(const float* bias, int ld, int Nq, int Nk_pad, float* out, hipStream_t s) {
    BG_REQUIRE(Nk_pad % SKT == 0 && Nk_pad > 0 && ld % 4 == 0 && ld >= Nk_pad, "pack_attn_bias: Nk_pad=%d ld=%d", Nk_pad, ld);
    const long total4 = attn_bias_packed_floats(Nq, Nk_pad) / 4;
    hipLaunchKernelGGL(pack_attn_bias_kernel, dim3((unsigned)cdiv(total4, 256L)), dim3(256), 0, s, bias, ld, Nq, Nk_pad / SKT, out, total4);
    LAUNCH_CHECK();
}

__device__ float g_zero_block[1024];   // stands in for an absent bias (zero-initialised device global; one copy per device)
static const float* zero_block() {
    static const float* ptr[64] = {};
    int dev = 0;
    HIP_CHECK(hipGetDevice(&dev));
    if (!ptr[dev]) HIP_CHECK(hipGetSymbolAddress((void**)&ptr[dev], HIP_SYMBOL(g_zero_block)));
    return ptr[dev];
}

#ifdef BEVGEN_ATTN_LAB
}  // namespace bevgen
#include "../../tools/attn_lab/attention_split_r1.inc"
namespace bevgen {
int g_attn_variant = 1;   // 0 = the round-1 kernel, 1 = this file's kernel, 2 = the same with phase stamps
void attn_lab_read_trace(unsigned long long* out) { HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_attn_trace), sizeof(g_attn_trace))); }
#endif

// merge of the key ranges of a split launch: one thread per (batch, head, query row, 4 output columns); ranges in order (deterministic)
__global__ __launch_bounds__(256) void attention_split_combine_kernel(AttnSplitArgs a) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)a.B * a.H * a.Nq * 16;
    if (i >= total) return;
    const int d = (int)(i & 15) * 4;
    const long row = i >> 4;   // (b * H + head) * Nq + q
    const int q = (int)(row % a.Nq), head = (int)((row / a.Nq) % a.H), b = (int)(row / ((long)a.Nq * a.H));
    const long stride = (long)a.B * a.H * a.Nq * kAttnPartPitch;
    const float* w0 = a.kws + row * kAttnPartPitch;
    float m = kNegBig;
    for (int k = 0; k < a.ksplit; ++k) m = fmaxf(m, w0[k * stride + 64]);
    float l = 0.f;
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < a.ksplit; ++k) {
        const float* w = w0 + k * stride;
        const float sc = __builtin_amdgcn_exp2f(w[64] - m);
        l = fmaf(w[65], sc, l);
        const float2 v0 = *reinterpret_cast<const float2*>(w + d), v1 = *reinterpret_cast<const float2*>(w + d + 2);
        o.x = fmaf(v0.x, sc, o.x); o.y = fmaf(v0.y, sc, o.y); o.z = fmaf(v1.x, sc, o.z); o.w = fmaf(v1.y, sc, o.w);
    }
    const float inv = 1.f / l;
    o.x *= inv; o.y *= inv; o.z *= inv; o.w *= inv;
    unsigned bad = 0;
    if (a.Op) store_planes4(a.Op + ((long)b * a.Nq + q) * 2 * (a.H * 64), head * 64 + d, o, bad);
    else *reinterpret_cast<float4*>(a.O + (long)b * a.o_bstride + (long)q * a.o_qstride + (long)head * a.o_hstride + d) = o;
    if (bad) status_raise(a.status, BG_ST_F16_RANGE);
}

void launch_attention_split(const AttnSplitArgs& a0, hipStream_t s) {
    AttnSplitArgs a = a0;
    a.status = status_current();
    if (a.ksplit < 1) a.ksplit = 1;
    BG_REQUIRE(a.ksplit == 1 || (a.kws && a.Nk_pad / SKT >= a.ksplit), "attention_split: key split %d needs a workspace and at least one key tile per range", a.ksplit);
    BG_REQUIRE(a.Nk_pad % SKT == 0 && a.Nk_pad > 0, "attention_split: Nk_pad=%d must be a positive multiple of %d", a.Nk_pad, SKT);
    a.bias_pk_tile_step = 1024;
    a.bias_pk_qb_stride = (long)(a.Nk_pad / SKT) * 1024;
    if (a.bias_pk == nullptr) {
        a.bias_pk = zero_block();
        a.bias_head_stride = 0; a.bias_pk_tile_step = 0; a.bias_pk_qb_stride = 0;
    }
    const dim3 grid(cdiv(a.Nq, 256), a.H, a.B * a.ksplit);
    ProfScope prof(PROF_ATTN, 4.0 * a.B * a.H * (double)a.Nq * a.Nk_pad * 64, s);
#ifdef BEVGEN_ATTN_LAB
    if (g_attn_variant == 0) launch_attention_split_r1(a, s);
    else if (g_attn_variant == 2) hipLaunchKernelGGL(attention_split_kernel<true>, grid, dim3(512), 0, s, a);
    else
#endif
    {
        if (a.ksplit > 1) hipLaunchKernelGGL((attention_split_kernel<false, true>), grid, dim3(512), 0, s, a);
        else hipLaunchKernelGGL((attention_split_kernel<false, false>), grid, dim3(512), 0, s, a);
    }
    LAUNCH_CHECK();
    if (a.ksplit > 1) {
        const long total = (long)a.B * a.H * a.Nq * 16;
        hipLaunchKernelGGL(attention_split_combine_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a);
        LAUNCH_CHECK();
    }
}

// ------------------------------------------------------------------------------------------------ operator-level entry (bevgen_op_attention in a split-precision context)
// fp32 q [B,H,Nq,64], k / v [B,H,Nk_pad,64] -> the kernel's operand images: Q planes with qmul (= score scale x log2 e) folded in, K planes, V^T planes
__global__ __launch_bounds__(256) void attn_split_operands_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v, _Float16* __restrict__ Qh,
                                                                 _Float16* __restrict__ Ql, _Float16* __restrict__ Kh, _Float16* __restrict__ Kl, _Float16* __restrict__ VTh,
                                                                 _Float16* __restrict__ VTl, long nq, long nk, int Nk_pad, float qmul, unsigned* __restrict__ status) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    unsigned bad = 0;
    if (i < nq) {
        const float x = q[i] * qmul;
        const _Float16 hi = (_Float16)x;
        guard_half(hi, bad);
        Qh[i] = hi; Ql[i] = (_Float16)((x - (float)hi) * kLo);
    }
    if (i < nk) {
        const float x = k[i];
        const _Float16 hi = (_Float16)x;
        guard_half(hi, bad);
        Kh[i] = hi; Kl[i] = (_Float16)((x - (float)hi) * kLo);
        const long bh = i / ((long)Nk_pad * 64), r = i % ((long)Nk_pad * 64);
        const int j = (int)(r / 64), d = (int)(r % 64);
        const float y = v[i];
        const _Float16 vh = (_Float16)y;
        guard_half(vh, bad);
        const long dst = bh * (long)Nk_pad * 64 + (long)d * Nk_pad + j;
        VTh[dst] = vh; VTl[dst] = (_Float16)((y - (float)vh) * kLo);
    }
    if (bad) status_raise(status, BG_ST_F16_RANGE);
}
void launch_attn_split_operands(const float* q, const float* k, const float* v, void* Qh, void* Ql, void* Kh, void* Kl, void* VTh, void* VTl, int B, int H, int Nq, int Nk_pad,
                                float qmul, hipStream_t s) {
    const long nq = (long)B * H * Nq * 64, nk = (long)B * H * Nk_pad * 64;
    hipLaunchKernelGGL(attn_split_operands_kernel, dim3((unsigned)cdiv(std::max(nq, nk), 256L)), dim3(256), 0, s, q, k, v, reinterpret_cast<_Float16*>(Qh), reinterpret_cast<_Float16*>(Ql),
                       reinterpret_cast<_Float16*>(Kh), reinterpret_cast<_Float16*>(Kl), reinterpret_cast<_Float16*>(VTh), reinterpret_cast<_Float16*>(VTl), nq, nk, Nk_pad, qmul, status_current());
    LAUNCH_CHECK();
}

// ------------------------------------------------------------------------------------------------ q / kv preparation with split outputs
// (muse_maskgit_pytorch.py:132-146: x8, null-kv concat, l2norm eps 1e-12, q_scale / k_scale), then hi/lo split; V is written transposed.
__device__ __forceinline__ void split1(float v, _Float16& hi, _Float16& lo, unsigned& bad) {
    hi = (_Float16)v;
    lo = (_Float16)((v - (float)hi) * kLo);
    guard_half(hi, bad);
}

__global__ __launch_bounds__(256) void muse_q_prep_split_kernel(const float* __restrict__ qraw, const float* __restrict__ q_scale, _Float16* __restrict__ Qh,
                                                                _Float16* __restrict__ Ql, int H, int Nq, long total, float post, unsigned* __restrict__ status) {
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
    unsigned bad = 0;
    split1(q, hi, lo, bad);
    const long dst = ((b * H + h) * Nq + n) * 64 + lane;
    Qh[dst] = hi;
    Ql[dst] = lo;
    if (bad) status_raise(status, BG_ST_F16_RANGE);
}

void launch_muse_q_prep_split(const float* qraw, const float* q_scale, void* Qh, void* Ql, int B, int H, int Nq, float post, hipStream_t s) {
    const long total = (long)B * Nq * H;
    hipLaunchKernelGGL(muse_q_prep_split_kernel, dim3((int)((total + 3) / 4)), dim3(256), 0, s, qraw, q_scale, reinterpret_cast<_Float16*>(Qh),
                       reinterpret_cast<_Float16*>(Ql), H, Nq, total, post, status_current());
    LAUNCH_CHECK();
}

// one workgroup = 64 consecutive key rows of one (batch, head): K rows are written row-major, V goes through LDS so that the transposed
// [64 dims][keys] image is written with contiguous 128-byte segments
__global__ __launch_bounds__(256) void muse_kv_prep_split_kernel(const float* __restrict__ kvraw, const float* __restrict__ null_kv, const float* __restrict__ k_scale,
                                                                 _Float16* __restrict__ Kh, _Float16* __restrict__ Kl, _Float16* __restrict__ VTh,
                                                                 _Float16* __restrict__ VTl, int H, int Nk, int Nk_pad, unsigned* __restrict__ status) {
    __shared__ float vt[64][65];
    unsigned bad = 0;
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
            split1(k, hi, lo, bad);
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
            split1(vt[lane][d], hi, lo, bad);
            VTh[kbase + (long)d * Nk_pad + j] = hi;
            VTl[kbase + (long)d * Nk_pad + j] = lo;
        }
    }
    if (bad) status_raise(status, BG_ST_F16_RANGE);
}

// prepared null key / value of one attention module for the fused to_kv epilogue (EPI_MUSE_KV): out = [k_hi | k_lo | v_hi | v_lo][H][64] halves
__global__ __launch_bounds__(64) void muse_null_kv_prep_kernel(const float* __restrict__ null_kv, const float* __restrict__ k_scale, _Float16* __restrict__ out, int H,
                                                               unsigned* __restrict__ status) {
    const int h = blockIdx.x, lane = threadIdx.x;
    const float kv = null_kv[h * 64 + lane], vv = null_kv[(long)H * 64 + h * 64 + lane];
    const float nrm = fmaxf(sqrtf(wave_sum(kv * kv)), 1e-12f);
    const float k = (kv / nrm) * k_scale[lane];
    _Float16 hi, lo;
    unsigned bad = 0;
    split1(k, hi, lo, bad);
    out[h * 64 + lane] = hi;
    out[H * 64 + h * 64 + lane] = lo;
    split1(vv, hi, lo, bad);
    out[2 * H * 64 + h * 64 + lane] = hi;
    out[3 * H * 64 + h * 64 + lane] = lo;
    if (bad) status_raise(status, BG_ST_F16_RANGE);
}
void launch_muse_null_kv_prep(const float* null_kv, const float* k_scale, void* out, int H, hipStream_t s) {
    hipLaunchKernelGGL(muse_null_kv_prep_kernel, dim3(H), dim3(64), 0, s, null_kv, k_scale, reinterpret_cast<_Float16*>(out), H, status_current());
    LAUNCH_CHECK();
}

void launch_muse_kv_prep_split(const float* kvraw, const float* null_kv, const float* k_scale, void* Kh, void* Kl, void* VTh, void* VTl, int B, int H, int Nk,
                               int Nk_pad, hipStream_t s) {
    dim3 grid(cdiv(Nk_pad, 64), H, B);
    hipLaunchKernelGGL(muse_kv_prep_split_kernel, grid, dim3(256), 0, s, kvraw, null_kv, k_scale, reinterpret_cast<_Float16*>(Kh), reinterpret_cast<_Float16*>(Kl),
                       reinterpret_cast<_Float16*>(VTh), reinterpret_cast<_Float16*>(VTl), H, Nk, Nk_pad, status_current());
    LAUNCH_CHECK();
}

}  // namespace bevgen
