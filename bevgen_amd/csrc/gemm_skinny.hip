// Skinny fp32 GEMM for the per-token decode step (Route A):  C[M,N] = A[M,K] * W[N,K]^T (+bias, GELU, residual), M <= 64.
//
// The step is weight-streaming bound (every W row is read exactly once per step; 1.1 GB of fp32 weights per step for the 24-layer
// model), so the kernel is organised around the W stream, not around an LDS tile:
//   * a workgroup owns 16 output columns (=> N/16 workgroups: 192 for the fused QKV, 256 for the MLP up-projection, enough to
//     pull on every CU's memory pipe) and its 8 waves split the workgroup's K range eight ways;
//   * each wave streams its W slice straight from HBM into registers with non-temporal 16-byte loads (the operand is not shared
//     between waves, so an LDS round trip would be pure overhead), all loads of an unrolled group in flight before the first MFMA;
//   * the activations (<= 64 x K fp32, L2 resident) are the other operand of v_mfma_f32_16x16x4_f32 (exact fp32); the four k-slots of
//     one MFMA belong to the four 16-lane quarters, and quarter q is given the CONTIGUOUS k-range [16c+4q, 16c+4q+4) of every
//     16-wide chunk, so one 16-byte load per lane feeds 4 MFMAs (any k permutation is legal as long as A and W agree);
//   * partial sums of the 8 waves are reduced through LDS in a fixed order (deterministic); narrow layers (N/16 < 128 workgroups) are
//     additionally split over K across workgroups, partials to a workspace + a tiny fixed-order reduce kernel with the epilogue.
#include "common.h"
#include "kernels.h"
#include "profiler.h"

namespace bevgen {

constexpr int SK_WAVES = 8;
constexpr int SK_NT = 16;  // output columns per workgroup

__device__ __forceinline__ float4 ldg_nt(const float* p) {
    const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
    return make_float4(v[0], v[1], v[2], v[3]);
}

template <int MT>  // number of 16-row M tiles (M <= 16*MT)
__global__ __launch_bounds__(SK_WAVES * 64) void gemm_skinny_kernel(GemmArgs g, float* __restrict__ partial, int ksplit) {
    __shared__ float red[SK_WAVES - 1][MT][4][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int n0 = blockIdx.x * SK_NT;
    const int split = blockIdx.y;
    const int kper = g.K / (ksplit * SK_WAVES);  // per wave; multiple of 16 (checked by the launcher)
    const int kbeg = (split * SK_WAVES + wave) * kper;

    const int n = min(n0 + r, g.N - 1);
    const float* wp = g.B + (long)n * g.ldb + kbeg + 4 * q;
    const float* ap[MT];
    bool avalid[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int m = 16 * t + r;
        avalid[t] = m < g.M;
        ap[t] = g.A + (long)min(m, g.M - 1) * g.lda + kbeg + 4 * q;
    }

    f32x4 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    constexpr int UN = 8;  // 16-wide chunks per unrolled group: 8 x 1 KiB of W in flight per wave
    for (int k = 0; k < kper; k += 16 * UN) {
        float4 wv[UN], av[MT][UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const bool in = k + 16 * u < kper;
            wv[u] = in ? ldg_nt(wp + k + 16 * u) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int t = 0; t < MT; ++t)
                av[t][u] = (in && avalid[t]) ? *reinterpret_cast<const float4*>(ap[t] + k + 16 * u) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const float w4[4] = {wv[u].x, wv[u].y, wv[u].z, wv[u].w};
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                const float a4[4] = {av[t][u].x, av[t][u].y, av[t][u].z, av[t][u].w};
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[kk], w4[kk], acc[t], 0, 0, 0);
            }
        }
    }

    // cross-wave reduction, fixed order
    if (wave > 0) {
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) red[wave - 1][t][j][lane] = acc[t][j];
    }
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int w = 0; w < SK_WAVES - 1; ++w)
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[t][j] += red[w][t][j][lane];

    // C/D layout of the 16x16 MFMA: col = lane & 15, row = 4 * (lane >> 4) + reg
    const int col = n0 + r;
    if (col >= g.N) return;
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = 16 * t + 4 * q + j;
            if (m >= g.M) continue;
            if (ksplit > 1) {
                partial[((long)split * g.M + m) * g.N + col] = acc[t][j];
            } else {
                float v = acc[t][j] * g.alpha + (g.bias_n ? g.bias_n[col] : 0.f);
                if (g.act == ACT_GELU) v = gelu_erf(v);
                if (g.R) v += g.R[(long)m * g.ldr + col];
                g.C[(long)m * g.ldc + col] = v;
            }
        }
}

__global__ __launch_bounds__(256) void splitk_reduce_kernel(GemmArgs g, const float* __restrict__ partial, int ksplit) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)g.M * g.N) return;
    const int m = (int)(i / g.N), col = (int)(i % g.N);
    float s = 0.f;
    for (int k = 0; k < ksplit; ++k) s += partial[(long)k * g.M * g.N + i];
    float v = s * g.alpha + (g.bias_n ? g.bias_n[col] : 0.f);
    if (g.act == ACT_GELU) v = gelu_erf(v);
    if (g.R) v += g.R[(long)m * g.ldr + col];
    g.C[(long)m * g.ldc + col] = v;
}

void launch_splitk_reduce(const GemmArgs& g, const float* partial, int ksplit, hipStream_t s) {
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(cdiv((long)g.M * g.N, 256)), dim3(256), 0, s, g, partial, ksplit);
    LAUNCH_CHECK();
}

int gemm_skinny_ksplit(int M, int N, int K) {
    (void)M;
    const int blocks = cdiv(N, SK_NT);
    int s = 1;
    // split across workgroups only for narrow layers, and keep >= 64 k per wave
    while (blocks * s < 256 && K % (s * 2 * SK_WAVES * 16) == 0 && K / (s * 2 * SK_WAVES) >= 64) s *= 2;
    return s;
}

size_t gemm_skinny_ws_bytes(int M, int N, int K) { return (size_t)gemm_skinny_ksplit(M, N, K) * M * N * sizeof(float); }

void launch_gemm_skinny_ws(const GemmArgs& g, float* ws, hipStream_t stream) {
    BG_REQUIRE(g.M >= 1 && g.M <= 64, "gemm_skinny: M=%d must be in [1,64]", g.M);
    BG_REQUIRE(g.K % (SK_WAVES * 16) == 0 && g.lda % 4 == 0 && g.ldb % 4 == 0, "gemm_skinny: K=%d must be a multiple of %d, strides multiples of 4", g.K, SK_WAVES * 16);
    BG_REQUIRE(g.batch == 1 && g.bias_m == nullptr, "gemm_skinny: batch/bias_m unsupported");
    const int ks = gemm_skinny_ksplit(g.M, g.N, g.K);
    dim3 grid(cdiv(g.N, SK_NT), ks);
    ProfScope prof(PROF_GEMM_SKINNY, ((double)g.N * g.K + (double)g.M * g.K + (double)g.M * g.N) * sizeof(float), stream);  // work = algorithmic bytes (W once + x + y)
    const int mt = cdiv(g.M, 16);
    switch (mt) {
        case 1: hipLaunchKernelGGL(gemm_skinny_kernel<1>, grid, dim3(SK_WAVES * 64), 0, stream, g, ws, ks); break;
        case 2: hipLaunchKernelGGL(gemm_skinny_kernel<2>, grid, dim3(SK_WAVES * 64), 0, stream, g, ws, ks); break;
        case 3: hipLaunchKernelGGL(gemm_skinny_kernel<3>, grid, dim3(SK_WAVES * 64), 0, stream, g, ws, ks); break;
        default: hipLaunchKernelGGL(gemm_skinny_kernel<4>, grid, dim3(SK_WAVES * 64), 0, stream, g, ws, ks); break;
    }
    LAUNCH_CHECK();
    if (ks > 1) {
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(cdiv((long)g.M * g.N, 256)), dim3(256), 0, stream, g, ws, ks);
        LAUNCH_CHECK();
    }
}

}  // namespace bevgen
