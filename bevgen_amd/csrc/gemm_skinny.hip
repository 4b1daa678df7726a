// Skinny fp32 GEMM for the per-token decode step (Route A):  C[M,N] = A[M,K] * W[N,K]^T (+bias, GELU, residual), M <= 64.
//
// The step is weight-streaming bound (every W row is read exactly once per step, 1.1 GB of fp32 weights per step for the
// 24-layer model), so the kernel is organised around the W stream instead of an LDS tile:
//   * a workgroup owns 32 output columns; its 4 waves split the K range of the workgroup four ways and each wave
//     streams its W slice straight from HBM into registers (16 B per lane per load, no LDS round trip - the operand is
//     not shared between waves), several loads in flight before the first MFMA consumes them;
//   * the activations (<= 64 x K, L2 resident) are the other MFMA operand; one W fragment feeds both 32-row M tiles;
//   * v_mfma_f32_32x32x2_f32 keeps the arithmetic exact fp32; the lane halves take contiguous k-ranges of every
//     8-wide chunk (same trick as gemm.hip) so each 16-byte load feeds 4 MFMAs;
//   * K is additionally split over workgroups (deterministic: partials go to a workspace and a second kernel reduces them
//     in a fixed order and applies the epilogue) so that narrow layers still occupy all 256 CUs.
#include "common.h"
#include "kernels.h"
#include "profiler.h"

namespace bevgen {

template <int MT>
__global__ __launch_bounds__(256) void gemm_skinny_kernel(GemmArgs g, float* __restrict__ partial, int ksplit) {
    __shared__ float red[3][MT][16][64];  // waves 1..3 -> wave 0
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int n0 = blockIdx.x * 32;
    const int split = blockIdx.y;
    const int kper = g.K / (ksplit * 4);  // per wave; multiple of 8 (checked by the launcher)
    const int kbeg = (split * 4 + wave) * kper;

    const int n = min(n0 + r, g.N - 1);
    const float* wp = g.B + (long)n * g.ldb + kbeg + 4 * h;
    const float* ap[MT];
    bool avalid[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int m = 32 * t + r;
        avalid[t] = m < g.M;
        ap[t] = g.A + (long)min(m, g.M - 1) * g.lda + kbeg + 4 * h;
    }

    f32x16 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[t][q] = 0.f;

    constexpr int UN = 4;  // 8-wide chunks per unrolled group
    for (int k = 0; k < kper; k += 8 * UN) {
        float4 wv[UN], av[MT][UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const bool in = k + 8 * u < kper;
            wv[u] = in ? *reinterpret_cast<const float4*>(wp + k + 8 * u) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int t = 0; t < MT; ++t)
                av[t][u] = (in && avalid[t]) ? *reinterpret_cast<const float4*>(ap[t] + k + 8 * u) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const float w4[4] = {wv[u].x, wv[u].y, wv[u].z, wv[u].w};
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                const float a4[4] = {av[t][u].x, av[t][u].y, av[t][u].z, av[t][u].w};
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[kk], w4[kk], acc[t], 0, 0, 0);
            }
        }
    }

    // cross-wave reduction (fixed order 0+1+2+3)
    if (wave > 0) {
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int q = 0; q < 16; ++q) red[wave - 1][t][q][lane] = acc[t][q];
    }
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[t][q] = ((acc[t][q] + red[0][t][q][lane]) + red[1][t][q][lane]) + red[2][t][q][lane];

    const int col = n0 + r;
    if (col >= g.N) return;
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int m = 32 * t + (q & 3) + 8 * (q >> 2) + 4 * h;
            if (m >= g.M) continue;
            if (ksplit > 1) {
                partial[((long)split * g.M + m) * g.N + col] = acc[t][q];
            } else {
                float v = acc[t][q] * g.alpha + (g.bias_n ? g.bias_n[col] : 0.f);
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

int gemm_skinny_ksplit(int M, int N, int K) {
    const int blocks = cdiv(N, 32);
    int s = 1;
    while (blocks * s < 512 && K % (s * 2 * 4 * 8) == 0 && K / (s * 2 * 4) >= 64) s *= 2;
    return s;
}

size_t gemm_skinny_ws_bytes(int M, int N, int K) { return (size_t)gemm_skinny_ksplit(M, N, K) * M * N * sizeof(float); }

void launch_gemm_skinny_ws(const GemmArgs& g, float* ws, hipStream_t stream) {
    BG_REQUIRE(g.M >= 1 && g.M <= 64, "gemm_skinny: M=%d must be in [1,64]", g.M);
    BG_REQUIRE(g.K % 32 == 0 && g.lda % 4 == 0 && g.ldb % 4 == 0, "gemm_skinny: K must be a multiple of 32, strides multiples of 4");
    BG_REQUIRE(g.batch == 1 && g.bias_m == nullptr, "gemm_skinny: batch/bias_m unsupported");
    const int ks = gemm_skinny_ksplit(g.M, g.N, g.K);
    dim3 grid(cdiv(g.N, 32), ks);
    ProfScope prof(PROF_GEMM_SKINNY, ((double)g.N * g.K + (double)g.M * g.K + (double)g.M * g.N) * sizeof(float), stream);  // work = algorithmic bytes (W once + x + y)
    if (g.M <= 32)
        hipLaunchKernelGGL(gemm_skinny_kernel<1>, grid, dim3(256), 0, stream, g, ws, ks);
    else
        hipLaunchKernelGGL(gemm_skinny_kernel<2>, grid, dim3(256), 0, stream, g, ws, ks);
    LAUNCH_CHECK();
    if (ks > 1) {
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(cdiv((long)g.M * g.N, 256)), dim3(256), 0, stream, g, ws, ks);
        LAUNCH_CHECK();
    }
}

static float* g_skinny_ws = nullptr;
static size_t g_skinny_ws_bytes = 0;

void launch_gemm_skinny(const GemmArgs& g, hipStream_t stream) {
    const size_t need = gemm_skinny_ws_bytes(g.M, g.N, g.K);
    if (need > g_skinny_ws_bytes) {
        if (g_skinny_ws) HIP_CHECK(hipFree(g_skinny_ws));
        HIP_CHECK(hipMalloc(&g_skinny_ws, need));
        g_skinny_ws_bytes = need;
    }
    launch_gemm_skinny_ws(g, g_skinny_ws, stream);
}

}  // namespace bevgen
