// Normalisation kernels: LayerNorm (Route A nn.LayerNorm, Route M gamma-only LayerNorm muse_net:62-69), fused GEGLU+LayerNorm
// (muse_net:71-88), GroupNorm(32, eps=1e-6)(+swish) of the VQGAN decoder (stage1/model.py:29-35), row softmax.
// All are HBM/L2-bandwidth bound: one wave per row, 16-byte accesses where alignment allows, statistics two-pass in fp32
// (GroupNorm partial sums in fp64, reduced in a fixed order -> run-to-run deterministic).
#include "common.h"
#include "kernels.h"

namespace bevgen {

// ------------------------------------------------------------------------------------------------ LayerNorm
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float* __restrict__ y, int ldy, int rows, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (long)row * ldx;
    float* yr = y + (long)row * ldy;
    float s = 0.f;
    for (int i = lane; i < D; i += 64) s += xr[i];
    const float mean = wave_sum(s) / (float)D;
    float v = 0.f;
    for (int i = lane; i < D; i += 64) { const float d = xr[i] - mean; v = fmaf(d, d, v); }
    const float rstd = rsqrtf(wave_sum(v) / (float)D + eps);
    for (int i = lane; i < D; i += 64) {
        float o = (xr[i] - mean) * rstd * gamma[i];
        if (beta) o += beta[i];
        yr[i] = o;
    }
    for (int i = D + lane; i < ldy; i += 64) yr[i] = 0.f;
}

// Register-resident variant (the one normally used): the row is loaded ONCE with 16-byte loads that are all in flight together,
// statistics and output come from registers.  A decode step calls LayerNorm 49 times on 16-64 rows, where the scalar three-pass
// kernel above is pure load latency (19 us per call measured); this one is a single round trip.
constexpr int LN_MAXV = 12;  // float4 per lane -> D <= 3072

// PLANES: y is the interleaved (hi, lo) f16 plane image [row][ldy/32][2][32] read by the split-precision GEMM (gemm_split.hip) instead of fp32
// MAXV: float4 per lane the instantiation holds (4: rows up to 1024 wide - 16 row registers instead of 48, eight waves per SIMD instead of five)
template <bool PLANES, int MAXV = LN_MAXV>
__global__ __launch_bounds__(256) void layernorm_vec_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float* __restrict__ y, int ldy, int rows, int D, float eps, unsigned* __restrict__ status) {
    // D need not be a multiple of 4 (Route M: LayerNorm over F = 2730 columns of a row padded to 2752): the last vector of a row is then partly
    // padding - read (the row storage is ldx >= round_up(D, 4) wide), excluded from the statistics, written as zero.
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float4* xr = reinterpret_cast<const float4*>(x + (long)row * ldx);
    const int nv = (D + 3) >> 2;
    float4 v[MAXV];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        const int i = lane + 64 * j;
        v[j] = i < nv ? xr[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        if (4 * i + 3 >= D) {   // boundary / padding vector: elements at columns >= D do not exist
            if (4 * i + 0 >= D) v[j].x = 0.f;
            if (4 * i + 1 >= D) v[j].y = 0.f;
            if (4 * i + 2 >= D) v[j].z = 0.f;
            v[j].w = 0.f;
        }
    }
#pragma unroll
    for (int j = 0; j < MAXV; ++j) s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        const int i = lane + 64 * j;
        if (i < nv) {
            const float a = v[j].x - mean, b = v[j].y - mean, c = v[j].z - mean, d = v[j].w - mean;
            if (4 * i + 3 < D) q += (a * a + b * b) + (c * c + d * d);
            else q += ((4 * i + 0 < D ? a * a : 0.f) + (4 * i + 1 < D ? b * b : 0.f)) + (4 * i + 2 < D ? c * c : 0.f);
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
    float4* yr = reinterpret_cast<float4*>(y + (long)row * ldy);
    const float4* g4 = reinterpret_cast<const float4*>(gamma);
    const float4* b4 = reinterpret_cast<const float4*>(beta);
    unsigned bad = 0;   // PLANES: an output outside the f16 operand range (or NaN) - raised once per thread at the end (BG_ST_F16_RANGE)
    if (PLANES && (D & 255) == 0 && ldy == D) {
        // Whole rows of full vectors (the transformer widths): neighbouring lanes trade halves so that every lane stores 16 contiguous bytes - the even lane the hi
        // parts of the pair's 8 columns, the odd lane the lo parts - and a wave's store instruction covers 1 KiB of whole 128-byte lines (the generic path below writes
        // 8 bytes of each plane per lane: two instructions that each touch half of every line).
        const bool odd = lane & 1;
        _Float16* prow = reinterpret_cast<_Float16*>(y) + (long)row * 2 * ldy;
#pragma unroll
        for (int j = 0; j < MAXV; ++j) {
            const int i = lane + 64 * j;
            if (64 * j < nv) {   // (wave-uniform: nv is a multiple of 64)
                const float4 g = g4[i];
                float4 o = make_float4((v[j].x - mean) * rstd * g.x, (v[j].y - mean) * rstd * g.y, (v[j].z - mean) * rstd * g.z, (v[j].w - mean) * rstd * g.w);
                if (beta) { const float4 bb = b4[i]; o.x += bb.x; o.y += bb.y; o.z += bb.z; o.w += bb.w; }
                half4_t h, l;
                h[0] = split_hi(o.x); h[1] = split_hi(o.y); h[2] = split_hi(o.z); h[3] = split_hi(o.w);
                l[0] = split_lo(o.x, h[0]); l[1] = split_lo(o.y, h[1]); l[2] = split_lo(o.z, h[2]); l[3] = split_lo(o.w, h[3]);
                guard_half4(h, bad);
                const uint2 hw = __builtin_bit_cast(uint2, h), lw = __builtin_bit_cast(uint2, l);
                const uint2 send = odd ? hw : lw;
                const uint2 recv = make_uint2((unsigned)__builtin_amdgcn_mov_dpp((int)send.x, 0xB1, 0xf, 0xf, true), (unsigned)__builtin_amdgcn_mov_dpp((int)send.y, 0xB1, 0xf, 0xf, true));
                const uint4 out = odd ? make_uint4(recv.x, recv.y, lw.x, lw.y) : make_uint4(hw.x, hw.y, recv.x, recv.y);
                const int c = 8 * (i >> 1);
                *reinterpret_cast<uint4*>(prow + (c >> 5) * 64 + (c & 31) + (odd ? 32 : 0)) = out;
            }
        }
        if (bad) status_raise(status, BG_ST_F16_RANGE);
        return;
    }
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        const int i = lane + 64 * j;
        if (i < nv) {
            float4 o;
            if (4 * i + 3 < D) {
                const float4 g = g4[i];
                o = make_float4((v[j].x - mean) * rstd * g.x, (v[j].y - mean) * rstd * g.y, (v[j].z - mean) * rstd * g.z, (v[j].w - mean) * rstd * g.w);
                if (beta) { const float4 bb = b4[i]; o.x += bb.x; o.y += bb.y; o.z += bb.z; o.w += bb.w; }
            } else {   // boundary vector: element-wise, gamma / beta are only D long
                const float in[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
                float out[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int col = 4 * i + e;
                    out[e] = col < D ? (in[e] - mean) * rstd * gamma[col] + (beta ? beta[col] : 0.f) : 0.f;
                }
                o = make_float4(out[0], out[1], out[2], out[3]);
            }
            if (PLANES) store_planes4(reinterpret_cast<_Float16*>(y) + (long)row * 2 * ldy, i * 4, o, bad);
            else yr[i] = o;
        } else if (i < (ldy >> 2)) {
            if (PLANES) store_planes4(reinterpret_cast<_Float16*>(y) + (long)row * 2 * ldy, i * 4, make_float4(0.f, 0.f, 0.f, 0.f), bad);
            else yr[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    if (PLANES && bad) status_raise(status, BG_ST_F16_RANGE);
}

void launch_layernorm(const float* x, int ldx, const float* gamma, const float* beta, float* y, int ldy, int rows, int D, float eps, hipStream_t s) {
    if (rows <= 0) return;
    const bool vec = ldx % 4 == 0 && ldy % 4 == 0 && ldx >= round_up(D, 4) && ldy >= round_up(D, 4) && D <= 256 * LN_MAXV && ldy <= 256 * LN_MAXV &&
                     (reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(beta)) % 16 == 0;
    if (vec && D <= 1024 && ldy <= 1024)
        hipLaunchKernelGGL((layernorm_vec_kernel<false, 4>), dim3(cdiv(rows, 4)), dim3(256), 0, s, x, ldx, gamma, beta, y, ldy, rows, D, eps, (unsigned*)nullptr);
    else if (vec)
        hipLaunchKernelGGL(layernorm_vec_kernel<false>, dim3(cdiv(rows, 4)), dim3(256), 0, s, x, ldx, gamma, beta, y, ldy, rows, D, eps, (unsigned*)nullptr);
    else
        hipLaunchKernelGGL(layernorm_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, s, x, ldx, gamma, beta, y, ldy, rows, D, eps);
    LAUNCH_CHECK();
}

void launch_layernorm_planes(const float* x, int ldx, const float* gamma, const float* beta, void* planes, int ldy, int rows, int D, float eps, hipStream_t s) {
    if (rows <= 0) return;
    BG_REQUIRE(ldx % 4 == 0 && ldx >= round_up(D, 4) && ldy % 32 == 0 && ldy >= round_up(D, 4) && ldy <= 256 * LN_MAXV &&
                   (reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(planes) | reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(beta)) % 16 == 0,
               "layernorm_planes: unsupported shape D=%d ldx=%d ldy=%d", D, ldx, ldy);
    if (D <= 1024 && ldy <= 1024)
        hipLaunchKernelGGL((layernorm_vec_kernel<true, 4>), dim3(cdiv(rows, 4)), dim3(256), 0, s, x, ldx, gamma, beta, reinterpret_cast<float*>(planes), ldy, rows, D, eps, status_current());
    else
        hipLaunchKernelGGL(layernorm_vec_kernel<true>, dim3(cdiv(rows, 4)), dim3(256), 0, s, x, ldx, gamma, beta, reinterpret_cast<float*>(planes), ldy, rows, D, eps, status_current());
    LAUNCH_CHECK();
}

// Folded LayerNorm (GemmArgs::ln_*): the producing epilogue left (sum, sum of squares) per (32 columns, row); one thread per row adds its groups in fp64, in group order
// (consecutive threads read consecutive rows of a group: coalesced; deterministic), and leaves (mean, rstd) for the consuming projection's epilogue.
__global__ __launch_bounds__(256) void ln_stats_finalize_kernel(const float2* __restrict__ sums, float2* __restrict__ out, int rows, int groups, double inv_count, double eps) {
    // 64 rows per workgroup, four threads per row (the row's groups dealt round-robin, eight loads in flight each): 24 576 rows = 384 workgroups instead of 96, and a
    // thread's chain of dependent L2 round trips is a quarter as long.  The four partial sums are added in a fixed order: deterministic.
    __shared__ double sh[4][64][2];
    const int rl = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int m = min(blockIdx.x * 64 + rl, rows - 1);
    double s1 = 0.0, s2 = 0.0;
    for (int g0 = part; g0 < groups; g0 += 32) {
        float2 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = sums[(long)min(g0 + 4 * u, groups - 1) * rows + m];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (g0 + 4 * u < groups) { s1 += (double)v[u].x; s2 += (double)v[u].y; }
    }
    sh[part][rl][0] = s1; sh[part][rl][1] = s2;
    __syncthreads();
    if (part == 0 && blockIdx.x * 64 + rl < rows) {
        const double t1 = ((sh[0][rl][0] + sh[1][rl][0]) + sh[2][rl][0]) + sh[3][rl][0];
        const double t2 = ((sh[0][rl][1] + sh[1][rl][1]) + sh[2][rl][1]) + sh[3][rl][1];
        const double mean = t1 * inv_count;
        const double var = fmax(t2 * inv_count - mean * mean, 0.0);
        out[m] = make_float2((float)mean, (float)(1.0 / sqrt(var + eps)));
    }
}
void launch_ln_stats_finalize(const float* group_sums, float* row_stats, int rows, int groups, int count, float eps, hipStream_t s) {
    if (rows <= 0) return;
    hipLaunchKernelGGL(ln_stats_finalize_kernel, dim3(cdiv(rows, 64)), dim3(256), 0, s, reinterpret_cast<const float2*>(group_sums), reinterpret_cast<float2*>(row_stats), rows,
                       groups, 1.0 / (double)count, (double)eps);
    LAUNCH_CHECK();
}

// ------------------------------------------------------------------------------------------------ GEGLU + LayerNorm
constexpr int GEGLU_MAX_PER_LANE = 48;  // F <= 3072

template <bool PLANES>
__global__ __launch_bounds__(256) void geglu_layernorm_kernel(const float* __restrict__ h, int ldh, const float* __restrict__ gamma,
                                                              float* __restrict__ y, int ldy, int rows, int F, float eps, unsigned* __restrict__ status) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    unsigned bad = 0;
    const float* a = h + (long)row * ldh;
    const float* gate = a + F;
    float g[GEGLU_MAX_PER_LANE];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < GEGLU_MAX_PER_LANE; ++j) {
        const int i = lane + 64 * j;
        g[j] = 0.f;
        if (i < F) {
            g[j] = gate[i] * gelu_erf(a[i]);
            s += g[j];
        }
    }
    const float mean = wave_sum(s) / (float)F;
    float v = 0.f;
#pragma unroll
    for (int j = 0; j < GEGLU_MAX_PER_LANE; ++j) {
        const int i = lane + 64 * j;
        if (i < F) { const float d = g[j] - mean; v = fmaf(d, d, v); }
    }
    const float rstd = rsqrtf(wave_sum(v) / (float)F + eps);
    float* yr = y + (long)row * ldy;
#pragma unroll
    for (int j = 0; j < GEGLU_MAX_PER_LANE; ++j) {
        const int i = lane + 64 * j;
        if (i < ldy) {
            const float o = i < F ? (g[j] - mean) * rstd * gamma[i] : 0.f;
            if (PLANES) {
                _Float16* p = reinterpret_cast<_Float16*>(y) + (long)row * 2 * ldy + (i >> 5) * 64 + (i & 31);
                const _Float16 hi = split_hi(o);
                guard_half(hi, bad);
                p[0] = hi; p[32] = split_lo(o, hi);
            } else {
                yr[i] = o;
            }
        }
    }
    if (PLANES && bad) status_raise(status, BG_ST_F16_RANGE);
}

// 8-byte-vectorised variant for even F (F = 2730 at D = 1024): `a` and `gate` rows are 8-byte aligned, all loads of a row are issued
// together, half the load/store instructions of the scalar kernel.
constexpr int GEGLU_MAX2 = 24;  // float2 per lane -> F <= 3072

template <bool PLANES>
__global__ __launch_bounds__(256) void geglu_layernorm_vec2_kernel(const float* __restrict__ h, int ldh, const float* __restrict__ gamma, float* __restrict__ y,
                                                                   int ldy, int rows, int F, float eps, unsigned* __restrict__ status) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    unsigned bad = 0;
    const float2* a2 = reinterpret_cast<const float2*>(h + (long)row * ldh);
    const float2* g2 = reinterpret_cast<const float2*>(h + (long)row * ldh + F);
    const int n2 = F >> 1;
    float2 av[GEGLU_MAX2], gv[GEGLU_MAX2];
#pragma unroll
    for (int j = 0; j < GEGLU_MAX2; ++j) {
        const int i = lane + 64 * j;
        av[j] = i < n2 ? a2[i] : make_float2(0.f, 0.f);
        gv[j] = i < n2 ? g2[i] : make_float2(0.f, 0.f);
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < GEGLU_MAX2; ++j) {
        av[j].x = gv[j].x * gelu_erf(av[j].x);  // gate * gelu(x)  (x = first half, muse_net:74-76)
        av[j].y = gv[j].y * gelu_erf(av[j].y);
        s += av[j].x + av[j].y;
    }
    const float mean = wave_sum(s) / (float)F;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < GEGLU_MAX2; ++j) {
        if (lane + 64 * j < n2) {
            const float c = av[j].x - mean, d = av[j].y - mean;
            q += c * c + d * d;
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)F + eps);
    float2* yr = reinterpret_cast<float2*>(y + (long)row * ldy);
    const float2* gam = reinterpret_cast<const float2*>(gamma);
#pragma unroll
    for (int j = 0; j < GEGLU_MAX2; ++j) {
        const int i = lane + 64 * j;
        if (i < n2) {
            const float2 g = gam[i];
            const float2 o = make_float2((av[j].x - mean) * rstd * g.x, (av[j].y - mean) * rstd * g.y);
            if (PLANES) store_planes2(reinterpret_cast<_Float16*>(y) + (long)row * 2 * ldy, i * 2, o, bad);
            else yr[i] = o;
        } else if (i < (ldy >> 1)) {
            if (PLANES) store_planes2(reinterpret_cast<_Float16*>(y) + (long)row * 2 * ldy, i * 2, make_float2(0.f, 0.f), bad);
            else yr[i] = make_float2(0.f, 0.f);
        }
    }
    if (PLANES && bad) status_raise(status, BG_ST_F16_RANGE);
}

void launch_geglu_layernorm(const float* h, int ldh, const float* gamma, float* y, int ldy, int rows, int F, float eps, hipStream_t s) {
    BG_REQUIRE(F <= 64 * GEGLU_MAX_PER_LANE && ldy <= 64 * GEGLU_MAX_PER_LANE, "geglu_layernorm: inner width %d too large", F);
    if (rows <= 0) return;
    const bool vec = F % 2 == 0 && ldh % 2 == 0 && ldy % 2 == 0 && F <= 128 * GEGLU_MAX2 && ldy <= 128 * GEGLU_MAX2 &&
                     (reinterpret_cast<uintptr_t>(h) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(gamma)) % 8 == 0;
    if (vec)
        hipLaunchKernelGGL(geglu_layernorm_vec2_kernel<false>, dim3(cdiv(rows, 4)), dim3(256), 0, s, h, ldh, gamma, y, ldy, rows, F, eps, (unsigned*)nullptr);
    else
        hipLaunchKernelGGL(geglu_layernorm_kernel<false>, dim3(cdiv(rows, 4)), dim3(256), 0, s, h, ldh, gamma, y, ldy, rows, F, eps, (unsigned*)nullptr);
    LAUNCH_CHECK();
}

void launch_geglu_layernorm_planes(const float* h, int ldh, const float* gamma, void* planes, int ldy, int rows, int F, float eps, hipStream_t s) {
    if (rows <= 0) return;
    BG_REQUIRE(ldy % 32 == 0 && F <= 64 * GEGLU_MAX_PER_LANE && ldy <= 64 * GEGLU_MAX_PER_LANE, "geglu_layernorm_planes: unsupported shape F=%d ldy=%d", F, ldy);
    const bool vec = F % 2 == 0 && ldh % 2 == 0 && F <= 128 * GEGLU_MAX2 && ldy <= 128 * GEGLU_MAX2 &&
                     (reinterpret_cast<uintptr_t>(h) | reinterpret_cast<uintptr_t>(gamma)) % 8 == 0;
    if (vec)
        hipLaunchKernelGGL(geglu_layernorm_vec2_kernel<true>, dim3(cdiv(rows, 4)), dim3(256), 0, s, h, ldh, gamma, reinterpret_cast<float*>(planes), ldy, rows, F, eps, status_current());
    else
        hipLaunchKernelGGL(geglu_layernorm_kernel<true>, dim3(cdiv(rows, 4)), dim3(256), 0, s, h, ldh, gamma, reinterpret_cast<float*>(planes), ldy, rows, F, eps, status_current());
    LAUNCH_CHECK();
}

// ------------------------------------------------------------------------------------------------ GroupNorm (NHWC, 32 groups)
constexpr int GN_GROUPS = 32;
constexpr int GN_PIX_PER_BLOCK = 1024;

// stage 1: per (image, pixel chunk) partial (sum, sumsq) of every group, fp64, fixed reduction order
__global__ __launch_bounds__(256) void groupnorm_partial_kernel(const float* __restrict__ x, double* __restrict__ part, int hw, int C, int chunks) {
    __shared__ double sh[256][4][2];
    __shared__ double chs[1024][2];
    const int tid = threadIdx.x;
    const int chunk = blockIdx.x, n = blockIdx.y;
    const int q4 = C >> 2;                 // float4 per pixel
    const int cq = tid % q4, psub = tid / q4, pstep = 256 / q4;   // (q4 not a divisor of 256 - C = 96, 192, 384 -: the last 256 - pstep q4 threads sit out)
    const int p_begin = chunk * GN_PIX_PER_BLOCK;
    const int p_end = psub < pstep ? min(hw, p_begin + GN_PIX_PER_BLOCK) : 0;
    double s[4] = {0, 0, 0, 0}, ss[4] = {0, 0, 0, 0};
    const float* base = x + ((long)n * hw) * C + cq * 4;
    // four independent loads in flight per thread (one load per iteration left the pass latency bound: 0.59 ms per scene for ONE read of every activation, where the apply
    // pass reads and writes them in 0.7); the sums stay in the order p_begin + psub, + pstep, ... : bit-identical statistics
    int p = p_begin + psub;
    for (; p + 3 * pstep < p_end; p += 4 * pstep) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(base + (long)(p + u * pstep) * C);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            s[0] += v[u].x; ss[0] += (double)v[u].x * v[u].x;
            s[1] += v[u].y; ss[1] += (double)v[u].y * v[u].y;
            s[2] += v[u].z; ss[2] += (double)v[u].z * v[u].z;
            s[3] += v[u].w; ss[3] += (double)v[u].w * v[u].w;
        }
    }
    for (; p < p_end; p += pstep) {
        const float4 v = *reinterpret_cast<const float4*>(base + (long)p * C);
        s[0] += v.x; ss[0] += (double)v.x * v.x;
        s[1] += v.y; ss[1] += (double)v.y * v.y;
        s[2] += v.z; ss[2] += (double)v.z * v.z;
        s[3] += v.w; ss[3] += (double)v.w * v.w;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) { sh[tid][i][0] = s[i]; sh[tid][i][1] = ss[i]; }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {  // per-channel sums over the pixel sub-lanes, in order
        const int q = c >> 2, i = c & 3;
        double a = 0, b = 0;
        for (int ps = 0; ps < pstep; ++ps) { a += sh[ps * q4 + q][i][0]; b += sh[ps * q4 + q][i][1]; }
        chs[c][0] = a; chs[c][1] = b;
    }
    __syncthreads();
    if (tid < GN_GROUPS) {
        const int cpg = C / GN_GROUPS;
        double a = 0, b = 0;
        for (int c = tid * cpg; c < (tid + 1) * cpg; ++c) { a += chs[c][0]; b += chs[c][1]; }
        double* o = part + (((long)n * chunks + chunk) * GN_GROUPS + tid) * 2;
        o[0] = a; o[1] = b;
    }
}

__global__ void groupnorm_finalize_kernel(const double* __restrict__ part, float* __restrict__ stats, int chunks, double count, float eps, int total) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;  // n*32 + g
    if (i >= total) return;
    const int n = i / GN_GROUPS, g = i % GN_GROUPS;
    double a = 0, b = 0;
    for (int c = 0; c < chunks; ++c) {
        const double* p = part + (((long)n * chunks + c) * GN_GROUPS + g) * 2;
        a += p[0]; b += p[1];
    }
    const double mean = a / count;
    double var = b / count - mean * mean;
    if (var < 0) var = 0;
    stats[2 * i] = (float)mean;
    stats[2 * i + 1] = (float)(1.0 / sqrt(var + (double)eps));
}

size_t groupnorm_ws_bytes(int n, int hw) { return (size_t)n * cdiv(hw, GN_PIX_PER_BLOCK) * GN_GROUPS * 2 * sizeof(double); }

void launch_groupnorm_stats(const float* x, float* stats, void* ws, int n, int hw, int C, float eps, hipStream_t s) {
    BG_REQUIRE(C % GN_GROUPS == 0 && C % 4 == 0 && C <= 1024, "groupnorm: unsupported channel count %d", C);
    const int chunks = cdiv(hw, GN_PIX_PER_BLOCK);
    double* part = reinterpret_cast<double*>(ws);
    hipLaunchKernelGGL(groupnorm_partial_kernel, dim3(chunks, n), dim3(256), 0, s, x, part, hw, C, chunks);
    LAUNCH_CHECK();
    const int total = n * GN_GROUPS;
    hipLaunchKernelGGL(groupnorm_finalize_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, part, stats, chunks,
                       (double)hw * (C / GN_GROUPS), eps, total);
    LAUNCH_CHECK();
}

// Apply pass.  grid (blocks per image, n), 256 threads.  The channel count is a multiple of 128 in every VQGAN level (128 / 256 / 512): C / 4 float4 per pixel is a power of
// two and divides the block's 256 threads' stride, so a thread keeps ONE channel quad for its whole loop - its four (scale, shift) pairs a = rstd gamma, b = beta - mean a are
// formed once, and no index of the loop needs a division (round 4's flat grid-stride loop spent 64-bit divisions and four statistics / gamma / beta loads on every element:
// 4 TB/s where the LayerNorm writers reach 6).
// Statistics from the producing convolution's epilogue (GemmArgs::gn_part [n hw / 32][C / 4][2]: sum and sum of squares of 32 pixels x 4 channels, fp32): one workgroup per
// (group, image) adds its hw / 32 x cpg / 4 pairs in fp64, thread-strided then a fixed-order tree - run-to-run deterministic, no pass over the activation.
__global__ __launch_bounds__(256) void groupnorm_finalize_partials_kernel(const float* __restrict__ part, float* __restrict__ stats, int hw, int C, float eps) {
    __shared__ double sh[256][2];
    const int g = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
    const int cpg = C / GN_GROUPS, qpg = cpg >> 2, quads = C >> 2, rows = hw >> 5;
    const float2* p = reinterpret_cast<const float2*>(part) + ((long)n * rows) * quads + (long)g * qpg;
    double a = 0, b = 0;
    for (int e = tid; e < rows * qpg; e += 256) {
        const float2 v = p[(long)(e / qpg) * quads + (e % qpg)];
        a += v.x; b += v.y;
    }
    sh[tid][0] = a; sh[tid][1] = b;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) { sh[tid][0] += sh[tid + o][0]; sh[tid][1] += sh[tid + o][1]; }
        __syncthreads();
    }
    if (tid == 0) {
        const double count = (double)hw * cpg, mean = sh[0][0] / count;
        double var = sh[0][1] / count - mean * mean;
        if (var < 0) var = 0;
        stats[2 * ((long)n * GN_GROUPS + g)] = (float)mean;
        stats[2 * ((long)n * GN_GROUPS + g) + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
}
bool groupnorm_partials_supported(int hw, int C) { return hw % 256 == 0 && C % 128 == 0; }
size_t groupnorm_part_floats(int n, int hw, int C) { return (size_t)n * (hw / 32) * (C / 4) * 2; }
void launch_groupnorm_stats_from_partials(const float* part, float* stats, int n, int hw, int C, float eps, hipStream_t s) {
    BG_REQUIRE(groupnorm_partials_supported(hw, C), "groupnorm from partials: unsupported shape hw=%d C=%d", hw, C);
    hipLaunchKernelGGL(groupnorm_finalize_partials_kernel, dim3(GN_GROUPS, n), dim3(256), 0, s, part, stats, hw, C, eps);
    LAUNCH_CHECK();
}

template <bool PLANES>   // PLANES: y is the interleaved (hi, lo) f16 plane image [pixel][C/32][2][32] read by the LDS-DMA split-precision convolution
__global__ __launch_bounds__(256) void groupnorm_apply_kernel(const float* __restrict__ x, const float* __restrict__ stats, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float* __restrict__ y, int hw, int C, int do_swish, unsigned* __restrict__ status) {
    const int n = blockIdx.y;
    const int cpg = C / GN_GROUPS, q4 = C >> 2;
    const long per_img4 = (long)hw * q4;                           // float4 per image
    const int stride = (int)(gridDim.x * blockDim.x);             // a multiple of q4 (launcher)
    const int i0 = blockIdx.x * blockDim.x + threadIdx.x;
    const int cq = i0 & (q4 - 1);                                  // this thread's channel quad, for every iteration
    float sc[4], sh[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = cq * 4 + k;
        if (stats) {
            const float* st = stats + ((long)n * GN_GROUPS + c / cpg) * 2;
            sc[k] = st[1] * gamma[c];
            sh[k] = beta[c] - st[0] * sc[k];
        } else {   // no normalisation: the pass only re-lays the tensor out (launch_to_planes)
            sc[k] = 1.f; sh[k] = 0.f;
        }
    }
    const float4* xin = reinterpret_cast<const float4*>(x) + (long)n * per_img4;
    const int lg = 31 - __builtin_clz(q4);
    unsigned bad = 0;
    for (long i = i0; i < per_img4; i += stride) {
        const float4 v = xin[i];
        float out[4] = {fmaf(v.x, sc[0], sh[0]), fmaf(v.y, sc[1], sh[1]), fmaf(v.z, sc[2], sh[2]), fmaf(v.w, sc[3], sh[3])};
        if (do_swish) {
#pragma unroll
            for (int k = 0; k < 4; ++k) out[k] = out[k] / (1.f + expf(-out[k]));
        }
        if (PLANES) store_planes4(reinterpret_cast<_Float16*>(y) + ((long)n * hw + (i >> lg)) * 2 * C, cq * 4, make_float4(out[0], out[1], out[2], out[3]), bad);
        else reinterpret_cast<float4*>(y)[(long)n * per_img4 + i] = make_float4(out[0], out[1], out[2], out[3]);
    }
    if (PLANES && bad) status_raise(status, BG_ST_F16_RANGE);
}

// Any channel count with whole quads (C % 4 == 0, C % 32 == 0 groups): the channel quad of an element is found by a division per element.  Only reached where C / 4 is not a
// power of two (a VQGAN with ch_mult 3: C = 384) - the reference configurations (ch = 128, ch_mult 1 / 1 / 2 / 2 / 4) all take the kernel above.
template <bool PLANES>
__global__ __launch_bounds__(256) void groupnorm_apply_generic_kernel(const float* __restrict__ x, const float* __restrict__ stats, const float* __restrict__ gamma,
                                                                      const float* __restrict__ beta, float* __restrict__ y, int hw, int C, int do_swish,
                                                                      unsigned* __restrict__ status) {
    const int n = blockIdx.y;
    const int cpg = C / GN_GROUPS, q4 = C >> 2;
    const long per_img4 = (long)hw * q4;
    const float4* xin = reinterpret_cast<const float4*>(x) + (long)n * per_img4;
    unsigned bad = 0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < per_img4; i += (long)gridDim.x * blockDim.x) {
        const long pix = i / q4;
        const int cq = (int)(i - pix * q4);
        const float4 v = xin[i];
        const float in[4] = {v.x, v.y, v.z, v.w};
        float out[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = cq * 4 + k;
            float o = in[k];
            if (stats) {
                const float* st = stats + ((long)n * GN_GROUPS + c / cpg) * 2;
                const float sck = st[1] * gamma[c];
                o = fmaf(o, sck, beta[c] - st[0] * sck);   // (the same two roundings as the power-of-two kernel)
            }
            if (do_swish) o = o / (1.f + expf(-o));
            out[k] = o;
        }
        if (PLANES) store_planes4(reinterpret_cast<_Float16*>(y) + ((long)n * hw + pix) * 2 * C, cq * 4, make_float4(out[0], out[1], out[2], out[3]), bad);
        else reinterpret_cast<float4*>(y)[(long)n * per_img4 + i] = make_float4(out[0], out[1], out[2], out[3]);
    }
    if (PLANES && bad) status_raise(status, BG_ST_F16_RANGE);
}

static bool groupnorm_apply_pow2(int C) {
    const int q4 = C >> 2;
    return C % 4 == 0 && q4 > 0 && (q4 & (q4 - 1)) == 0 && (q4 <= 256 ? 256 % q4 == 0 : q4 % 256 == 0);
}
static dim3 groupnorm_apply_grid(int n, int hw, int C) {
    const int q4 = C >> 2;
    BG_REQUIRE(C % 4 == 0 && C % GN_GROUPS == 0, "groupnorm apply: C = %d must be a multiple of 32", C);
    const long per_img4 = (long)hw * q4;
    long bx = std::max<long>(1, std::min<long>((per_img4 + 255) / 256, std::max<long>(1, 4096 / std::max(n, 1))));
    if (groupnorm_apply_pow2(C) && q4 > 256) bx = std::max<long>(q4 / 256, bx / (q4 / 256) * (q4 / 256));   // stride = 256 bx must stay a multiple of q4
    return dim3((unsigned)bx, (unsigned)n);
}
template <bool PLANES>
static void groupnorm_apply_launch(const float* x, const float* stats, const float* gamma, const float* beta, float* y, int n, int hw, int C, int do_swish, hipStream_t s) {
    unsigned* st = PLANES ? status_current() : nullptr;
    if (groupnorm_apply_pow2(C)) hipLaunchKernelGGL(groupnorm_apply_kernel<PLANES>, groupnorm_apply_grid(n, hw, C), dim3(256), 0, s, x, stats, gamma, beta, y, hw, C, do_swish, st);
    else hipLaunchKernelGGL(groupnorm_apply_generic_kernel<PLANES>, groupnorm_apply_grid(n, hw, C), dim3(256), 0, s, x, stats, gamma, beta, y, hw, C, do_swish, st);
    LAUNCH_CHECK();
}

void launch_groupnorm_apply(const float* x, const float* stats, const float* gamma, const float* beta, float* y, int n, int hw, int C, int do_swish, hipStream_t s) {
    groupnorm_apply_launch<false>(x, stats, gamma, beta, y, n, hw, C, do_swish, s);
}

// fp32 NHWC activation -> the (hi, lo) plane image the LDS-DMA convolution reads, values unchanged (x 1 + 0 is exact)
void launch_to_planes(const float* x, void* planes, int n, int hw, int C, hipStream_t s) {
    BG_REQUIRE(C % 32 == 0, "to_planes: C=%d must be a multiple of 32", C);
    groupnorm_apply_launch<true>(x, nullptr, nullptr, nullptr, reinterpret_cast<float*>(planes), n, hw, C, 0, s);
}

void launch_groupnorm_apply_planes(const float* x, const float* stats, const float* gamma, const float* beta, void* planes, int n, int hw, int C, int do_swish, hipStream_t s) {
    BG_REQUIRE(C % 32 == 0, "groupnorm_apply_planes: C=%d must be a multiple of 32", C);
    groupnorm_apply_launch<true>(x, stats, gamma, beta, reinterpret_cast<float*>(planes), n, hw, C, do_swish, s);
}

// ------------------------------------------------------------------------------------------------ row softmax (VQGAN AttnBlock, s1model:179-181)
__global__ __launch_bounds__(256) void row_softmax_kernel(float* __restrict__ x, int rows, int cols, float scale, int ld) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float* xr = x + row * ld;
    float mx = kNegBig;
    for (int i = lane; i < cols; i += 64) mx = fmaxf(mx, xr[i] * scale);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int i = lane; i < cols; i += 64) sum += expf(xr[i] * scale - mx);
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
    for (int i = lane; i < cols; i += 64) xr[i] = expf(xr[i] * scale - mx) * inv;
    for (int i = cols + lane; i < ld; i += 64) xr[i] = 0.f;
}

void launch_row_softmax(float* x, int rows, int cols, float scale, hipStream_t s, int ld) {
    hipLaunchKernelGGL(row_softmax_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, s, x, rows, cols, scale, ld > 0 ? ld : cols);
    LAUNCH_CHECK();
}

// ------------------------------------------------------------------------------------------------ misc
__global__ void fill_kernel(float* p, long n, float v) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = v;
}
void launch_fill(float* p, long n, float v, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(fill_kernel, dim3((int)std::min<long>((n + 255) / 256, 4096)), dim3(256), 0, s, p, n, v);
    LAUNCH_CHECK();
}
__global__ void scale_kernel(float* p, long n, float v) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] *= v;
}
void launch_scale(float* p, long n, float v, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(scale_kernel, dim3((int)std::min<long>((n + 255) / 256, 4096)), dim3(256), 0, s, p, n, v);
    LAUNCH_CHECK();
}
__global__ void add_kernel(const float* a, const float* b, float* c, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) c[i] = a[i] + b[i];
}
void launch_add(const float* a, const float* b, float* c, long n, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(add_kernel, dim3((int)std::min<long>((n + 255) / 256, 4096)), dim3(256), 0, s, a, b, c, n);
    LAUNCH_CHECK();
}

}  // namespace bevgen
