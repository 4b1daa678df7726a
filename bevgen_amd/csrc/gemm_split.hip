// Split-precision GEMM: fp32-class accuracy on the f16 matrix cores.   C[M,N] = A[M,K] * B[N,K]^T (+bias, GELU, residual)
//
// CDNA4 has no TF32/xf32 mode and its fp32-input MFMA runs at 1/16 of the f16 rate (157 vs 2500 TFLOP/s).  Each fp32 operand is
// therefore split into two f16 numbers with a scaled residual,
//        x = hi + lo * 2^-11,      hi = f16(x),   lo = f16((x - f32(hi)) * 2^11)
// (22 mantissa bits kept; the 2^11 scale keeps `lo` out of the f16 subnormal range), and the product is evaluated as
//        A B^T  ~=  hi_a hi_b^T  +  2^-11 (hi_a lo_b^T + lo_a hi_b^T)
// i.e. THREE v_mfma_f32_32x32x16_f16 per 16-deep k-step instead of EIGHT v_mfma_f32_32x32x2_f32 of four times the latency: 5.3x
// less matrix-pipe time.  f16 x f16 products are exact in fp32, accumulation is fp32; the dropped lo*lo term and the rounding of `lo`
// are each 2^-22 relative per product, so the result carries ~4 fp32 ulps of extra error (measured in tests/test_ops_gpu.py).
// Range: hi overflows for |x| >= 65520 -> inf/NaN in the output (loud, not silent); every GEMM input on this path is a normalised
// activation (LayerNorm / GroupNorm+swish / softmax-weighted values) or a weight, orders of magnitude below that.
//
// Weights (the B operand) are split once in bevgen_finalize; activations (A, fp32 in HBM) are split in the tile loader on the VALU,
// which runs beside the matrix pipe.  Tile 128x128x32, 4 waves x (2x2) MFMA tiles, two accumulator sets (main / correction);
// LDS rows of 32 halves padded to 40 (80 B) so the 16-lane groups of ds_read_b128 touch 16 distinct bank quads.
// The implicit-GEMM 3x3 convolution mode (NHWC gather, fused nearest-2x upsample) is shared with gemm.hip.
#include "common.h"
#include "kernels.h"
#include "profiler.h"

namespace bevgen {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

constexpr int SBM = 128, SBN = 128, SBK = 32, SLD = 40;  // SLD = padded LDS row stride in halves
constexpr float kLoScale = 2048.f, kLoInv = 1.f / 2048.f;

__device__ __forceinline__ void split8(const float4& p, const float4& q, half8& hi, half8& lo) {
    const float v[8] = {p.x, p.y, p.z, p.w, q.x, q.y, q.z, q.w};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const _Float16 h = (_Float16)v[i];
        hi[i] = h;
        lo[i] = (_Float16)((v[i] - (float)h) * kLoScale);
    }
}

struct AFrag { float4 v[2][2]; };            // 2 chunks of 8 floats per thread
struct BFrag { uint4 hi[2], lo[2]; };        // 2 chunks of 8 halves per thread, both planes

template <int MODE>
__device__ __forceinline__ void load_a(const GemmArgs& g, const float* __restrict__ A, int m0, int k0, int tid, AFrag& r, const int (&rowinfo)[2][3]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int e = tid + 256 * i;
        const int row = e >> 2, c8 = e & 3;
        const int m = m0 + row;
        float4 p = make_float4(0.f, 0.f, 0.f, 0.f), q = p;
        const float* src = nullptr;
        if (m < g.M) {
            if (MODE == MODE_PLAIN) {
                src = A + (long)m * g.lda + k0 + c8 * 8;
            } else {
                const int tap = k0 / g.conv_cin;
                const int c0 = k0 - tap * g.conv_cin;
                const int kh = tap / 3, kw = tap - kh * 3;
                int yy = rowinfo[i][1] * g.conv_stride + kh - g.conv_pad, xx = rowinfo[i][2] * g.conv_stride + kw - g.conv_pad;
                const int lim_h = g.conv_up ? 2 * g.conv_hin : g.conv_hin, lim_w = g.conv_up ? 2 * g.conv_win : g.conv_win;
                if (yy >= 0 && yy < lim_h && xx >= 0 && xx < lim_w) {
                    if (g.conv_up) { yy >>= 1; xx >>= 1; }
                    src = A + (((long)rowinfo[i][0] * g.conv_hin + yy) * g.conv_win + xx) * g.conv_cin + c0 + c8 * 8;
                }
            }
        }
        if (src) {
            p = *reinterpret_cast<const float4*>(src);
            q = *reinterpret_cast<const float4*>(src + 4);
        }
        r.v[i][0] = p;
        r.v[i][1] = q;
    }
}

__device__ __forceinline__ void load_b(const GemmArgs& g, int n0, int k0, int tid, BFrag& r) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int e = tid + 256 * i;
        const int row = e >> 2, c8 = e & 3;
        const int n = n0 + row;
        uint4 h = make_uint4(0, 0, 0, 0), l = h;
        if (n < g.N) {
            const long off = ((long)n * g.ldb + k0) * 2 + c8 * 8;   // interleaved planes: (row, 32-k block) = 64 halves, lo = hi + 32
            h = *reinterpret_cast<const uint4*>(g.B_hi + off);
            l = *reinterpret_cast<const uint4*>(g.B_lo + off);
        }
        r.hi[i] = h;
        r.lo[i] = l;
    }
}

__device__ __forceinline__ void store_a(_Float16* __restrict__ Shi, _Float16* __restrict__ Slo, int tid, const AFrag& r, unsigned& bad) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int e = tid + 256 * i;
        const int row = e >> 2, c8 = e & 3;
        half8 hi, lo;
        split8(r.v[i][0], r.v[i][1], hi, lo);
        guard_half8(hi, bad);   // an activation outside the f16 operand range (BG_ST_F16_RANGE, raised once at the end of the kernel)
        *reinterpret_cast<half8*>(Shi + row * SLD + c8 * 8) = hi;
        *reinterpret_cast<half8*>(Slo + row * SLD + c8 * 8) = lo;
    }
}

__device__ __forceinline__ void store_b(_Float16* __restrict__ Shi, _Float16* __restrict__ Slo, int tid, const BFrag& r) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int e = tid + 256 * i;
        const int row = e >> 2, c8 = e & 3;
        *reinterpret_cast<uint4*>(Shi + row * SLD + c8 * 8) = r.hi[i];
        *reinterpret_cast<uint4*>(Slo + row * SLD + c8 * 8) = r.lo[i];
    }
}

template <int MODE>
__global__ __launch_bounds__(256, 2) void gemm_split_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) _Float16 smem_h[];
    // per stage: A_hi, A_lo, B_hi, B_lo, each [128][SLD]
    constexpr int PLANE = SBM * SLD;
    _Float16* stage0 = smem_h;
    _Float16* stage1 = smem_h + 4 * PLANE;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, h = lane >> 5;
    int tx, ty;
    xcd_tile(gridDim.x, gridDim.y, tx, ty);
    const int n0 = tx * SBN, m0 = ty * SBM;
    const int bz = blockIdx.z;
    const float* A = g.A + (long)bz * g.strideA;

    int rowinfo[2][3] = {};
    if (MODE == MODE_CONV3) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = m0 + ((tid + 256 * i) >> 2);
            const int hw = g.conv_h * g.conv_w;
            const int img = m / hw;
            const int rem = m - img * hw;
            rowinfo[i][0] = img;
            rowinfo[i][1] = rem / g.conv_w;
            rowinfo[i][2] = rem - rowinfo[i][1] * g.conv_w;
        }
    }

    f32x16 accM[2][2], accC[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) { accM[i][j][q] = 0.f; accC[i][j][q] = 0.f; }

    const int nk = g.K / SBK;
    AFrag ra;
    BFrag rb;
    load_a<MODE>(g, A, m0, 0, tid, ra, rowinfo);
    load_b(g, n0, 0, tid, rb);
    unsigned bad = 0;
    store_a(stage0, stage0 + PLANE, tid, ra, bad);
    store_b(stage0 + 2 * PLANE, stage0 + 3 * PLANE, tid, rb);
    __syncthreads();

    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        if (more) {
            load_a<MODE>(g, A, m0, (kt + 1) * SBK, tid, ra, rowinfo);
            load_b(g, n0, (kt + 1) * SBK, tid, rb);
        }
        const _Float16* st = cur ? stage1 : stage0;
        const _Float16* a_hi = st + (wm * 64 + r) * SLD + h * 8;
        const _Float16* a_lo = a_hi + PLANE;
        const _Float16* b_hi = st + 2 * PLANE + (wn * 64 + r) * SLD + h * 8;
        const _Float16* b_lo = b_hi + PLANE;
#pragma unroll
        for (int ks = 0; ks < SBK / 16; ++ks) {
            half8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ah[i] = *reinterpret_cast<const half8*>(a_hi + i * 32 * SLD + ks * 16);
                al[i] = *reinterpret_cast<const half8*>(a_lo + i * 32 * SLD + ks * 16);
                bh[i] = *reinterpret_cast<const half8*>(b_hi + i * 32 * SLD + ks * 16);
                bl[i] = *reinterpret_cast<const half8*>(b_lo + i * 32 * SLD + ks * 16);
            }
            // issue order: no accumulator is reused by the next MFMA (a dependent back-to-back pair stalls the matrix pipe for the
            // full result latency) - every accumulator is touched again only 4 instructions later
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) accM[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], accM[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) accC[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], accC[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) accC[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], accC[i][j], 0, 0, 0);
        }
        if (more) {
            _Float16* nx = cur ? stage0 : stage1;
            store_a(nx, nx + PLANE, tid, ra, bad);
            store_b(nx + 2 * PLANE, nx + 3 * PLANE, tid, rb);
        }
        __syncthreads();
        cur ^= 1;
    }

    if (bad) status_raise(g.status, BG_ST_F16_RANGE);
    float* C = g.C + (long)bz * g.strideC;
    const float* R = g.R ? g.R + (long)bz * g.strideR : nullptr;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 64 + j * 32 + r;
            if (n >= g.N) continue;
            const float bn = g.bias_n ? g.bias_n[n] : 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int m = m0 + wm * 64 + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
                if (m >= g.M) continue;
                float v = (accM[i][j][q] + accC[i][j][q] * kLoInv) * g.alpha + bn;
                if (g.bias_m) v += g.bias_m[m];
                if (g.act == ACT_GELU) v = gelu_erf(v);
                if (R) v += R[(long)m * g.ldr + n];
                C[(long)m * g.ldc + n] = v;
            }
        }
}

void launch_gemm_split(const GemmArgs& g_in, hipStream_t stream) {
    GemmArgs g = g_in;
    if (g.mode == MODE_CONV3) {
        if (g.conv_stride == 0) g.conv_stride = 1;
        if (g.conv_pad < 0) g.conv_pad = 1;
        if (g.conv_hin == 0) g.conv_hin = g.conv_up ? g.conv_h / 2 : g.conv_h;
        if (g.conv_win == 0) g.conv_win = g.conv_up ? g.conv_w / 2 : g.conv_w;
    }
    BG_REQUIRE(g.B_hi && g.B_lo, "gemm_split: the B operand has not been split");
    g.status = status_current();
    BG_REQUIRE(g.K % SBK == 0 && g.lda % 4 == 0 && g.ldb % 8 == 0, "gemm_split: K %% 32, lda %% 4, ldb %% 8 required (K=%d lda=%d ldb=%d)", g.K, g.lda, g.ldb);
    BG_REQUIRE(g.batch == 1 || g.strideB == 0, "gemm_split: batched B operands are not split");
    if (g.mode == MODE_CONV3) BG_REQUIRE(g.conv_cin % SBK == 0 && g.K == 9 * g.conv_cin, "conv3x3: Cin=%d must be a multiple of 32", g.conv_cin);
    dim3 grid(cdiv(g.N, SBN), cdiv(g.M, SBM), g.batch);
    const size_t lds = (size_t)2 * 4 * SBM * SLD * sizeof(_Float16);  // 80 KiB
    static std::atomic<bool> attr_set[kMaxDevices];
    const int dslot = device_slot();
    if (!attr_set[dslot].load(std::memory_order_acquire)) {
        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_split_kernel<MODE_PLAIN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_split_kernel<MODE_CONV3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set[dslot].store(true, std::memory_order_release);
    }
    ProfScope prof(g.mode == MODE_CONV3 ? PROF_CONV3 : PROF_GEMM, 2.0 * g.M * (double)g.N * g.K * g.batch, stream);
    if (g.mode == MODE_CONV3)
        hipLaunchKernelGGL(gemm_split_kernel<MODE_CONV3>, grid, dim3(256), lds, stream, g);
    else
        hipLaunchKernelGGL(gemm_split_kernel<MODE_PLAIN>, grid, dim3(256), lds, stream, g);
    LAUNCH_CHECK();
}

// Split of an fp32 matrix (row length a multiple of 32) into the INTERLEAVED plane layout every split-precision kernel reads:
//     planes[row][k/32][0][k%32] = hi,  planes[row][k/32][1][k%32] = lo          (2 halves per element, 128 bytes per (row, 32-k block))
// so that the 32 hi and 32 lo values of one k-block share one 128-byte line: the LDS-DMA GEMM moves whole lines (a 64-byte half-line
// request still costs a full line of L2->L1 bandwidth, measured) and a row needs one DMA piece instead of two.
__global__ void split_weight_kernel(const float* __restrict__ w, _Float16* __restrict__ planes, long n, unsigned* __restrict__ status) {
    unsigned bad = 0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float v = w[i];
        const _Float16 h = (_Float16)v;
        guard_half(h, bad);
        const long o = (i >> 5) * 64 + (i & 31);
        planes[o] = h;
        planes[o + 32] = (_Float16)((v - (float)h) * kLoScale);
    }
    if (bad) status_raise(status, BG_ST_F16_RANGE);   // (weights: bevgen_finalize synchronises and reports it)
}

void launch_split_weight(const float* w, void* planes, long n, hipStream_t s) {
    BG_REQUIRE(n % 32 == 0, "split_weight: element count %ld must be a multiple of 32", n);
    hipLaunchKernelGGL(split_weight_kernel, dim3((int)std::min<long>((n + 255) / 256, 8192)), dim3(256), 0, s, w, reinterpret_cast<_Float16*>(planes), n, status_current());
    LAUNCH_CHECK();
}

}  // namespace bevgen
