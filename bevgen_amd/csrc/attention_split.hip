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
    const float* Bp = a.bias ? a.bias + (long)head * a.bias_head_stride + (long)qc * a.ldbias + 4 * h : nullptr;

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
        const long vo = koff + (long)vr * a.Nk_pad + tile * SKT + vc * 8;   // V^T [.., 64, Nk_pad]: same number of elements per (b,h) as K
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
    // bias row segment of a key tile (this lane's 16 keys), fetched one tile ahead: an L2 round trip is longer than the QK^T MFMAs in front of it
    float4 bv[4];
    auto gload_bias = [&](int tile) {
#pragma unroll
        for (int g = 0; g < 4; ++g) bv[g] = Bp ? *reinterpret_cast<const float4*>(Bp + tile * SKT + 8 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
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
        {   // first k-step: accumulate onto the inline constant 0 (no 32 v_mov per tile to clear the accumulators)
            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            const half8 ah = *reinterpret_cast<const half8*>(kh);
            const half8 al = *reinterpret_cast<const half8*>(kl);
            sC = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, ql[0], zero, 0, 0, 0);
            sM = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, qh[0], zero, 0, 0, 0);
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

        // ---- scale + bias, online softmax
        float sv[16];
        float mx = kNegBig;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float bb[4] = {bv[g].x, bv[g].y, bv[g].z, bv[g].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                sv[4 * g + j] = fmaf(sC[4 * g + j], kLoI, sM[4 * g + j]) + bb[j];   // the score scale lives in the Q planes
                mx = fmaxf(mx, sv[4 * g + j]);
            }
        }
        if (more) gload_bias(tile + 1);
        mx = fmaxf(mx, xor32(mx));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        float psum = 0.f;
        half8 ph[2], pl[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float p = __builtin_amdgcn_exp2f(sv[r] - m_new);   // scores are in the base-2 domain (scale, bias pre-multiplied by log2 e)
            psum += p;
            // p in [0,1] = hi + lo with lo stored UNSCALED: |lo| <= 2^-12, and what the f16 subnormal range drops is below 2^-25 ABSOLUTE, which is
            // what matters for a probability (a weight of the row sum); saves the 2^11 scaling and lets V_hi P_lo accumulate straight into oM
            const _Float16 hi = (_Float16)p;
            ph[r >> 3][r & 7] = hi;
            pl[r >> 3][r & 7] = (_Float16)(p - (float)hi);
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

        // ---- O^T += V^T P^T.  k-step s covers keys [16s, 16s+16); lane half h owns keys 16s + {0..3} + 4h and 16s + 8 + {0..3} + 4h
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

void launch_attention_split(const AttnSplitArgs& a, hipStream_t s) {
    BG_REQUIRE(a.Nk_pad % SKT == 0 && a.Nk_pad > 0, "attention_split: Nk_pad=%d must be a positive multiple of %d", a.Nk_pad, SKT);
    BG_REQUIRE(a.bias == nullptr || a.ldbias % 4 == 0, "attention_split: bias row stride must be a multiple of 4");
    dim3 grid(cdiv(a.Nq, 128), a.H, a.B);
    ProfScope prof(PROF_ATTN, 4.0 * a.B * a.H * (double)a.Nq * a.Nk_pad * 64, s);
    hipLaunchKernelGGL(attention_split_kernel, grid, dim3(256), 0, s, a);
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
