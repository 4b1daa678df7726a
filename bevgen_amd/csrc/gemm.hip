// fp32 NT GEMM on the CDNA4 matrix cores:  C[M,N] = A[M,K] * B[N,K]^T  (+bias, activation, residual)
//
// Both operands are K-contiguous (nn.Linear weight layout [out,in]; activations [rows,features]), so a block
// tile is staged with straight 16-byte row-segment copies.  Arithmetic is v_mfma_f32_32x32x2_f32: exact fp32
// products/accumulation (an fmaf chain), 64 FLOP/clk/SIMD = 157 TFLOP/s chip peak (MI355X_MICROARCH.md).
//
// Tile: 128x128x32 per 256-thread workgroup (4 waves as 2x2, each wave 64x64 = 2x2 MFMA tiles of 32x32).
// K-ordering trick: the two k-slots of one 32x32x2 MFMA are fed by lane halves (lane>>5).  Any permutation of k
// is legal as long as A and B agree, so half h is given the CONTIGUOUS k-range [8c+4h, 8c+4h+4) of every 8-wide
// chunk: one ds_read_b128 per operand row then feeds 4 consecutive MFMAs.  LDS rows are padded to 36 floats
// (144 B) so the 16-lane groups of ds_read_b128 hit 16 distinct bank quads (conflict-free).
//
// The same main loop serves the VQGAN decoder's 3x3 convolutions as an implicit GEMM (MODE_CONV3): the A-tile
// loader gathers NHWC input rows for tap (kh,kw) (zero at the border), optionally through a fused nearest-2x
// upsample (Upsample.forward, stage1/model.py:49-53).  Cin is a multiple of 32, so every 32-wide k-chunk lies
// inside one tap.
#include "common.h"
#include "kernels.h"
#include "profiler.h"
#include <unordered_map>

namespace bevgen {

constexpr int BM = 128, BN = 128, BK = 32, LDSS = 36;  // LDSS = padded LDS row stride (floats)

struct Frag4 { float4 v[4]; };

template <int MODE>
__device__ __forceinline__ void load_a_tile(const GemmArgs& g, const float* __restrict__ A, int m0, int k0, int tid, Frag4& r,
                                            const int (&rowinfo)[4][3]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = tid + 256 * i;
        const int row = e >> 3, c4 = e & 7;
        const int m = m0 + row;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (MODE == MODE_PLAIN) {
            if (m < g.M) v = *reinterpret_cast<const float4*>(A + (long)m * g.lda + k0 + c4 * 4);
        } else {  // implicit im2col, NHWC input
            if (m < g.M) {
                const int tap = k0 / g.conv_cin;
                const int c0 = k0 - tap * g.conv_cin;
                const int kh = tap / 3, kw = tap - kh * 3;
                // input coordinate of this tap: stride 1|2, zero padding `conv_pad` on the top/left (the bottom/right padding is implied by
                // the bounds test), optional nearest-2x upsample of the stored input
                int yy = rowinfo[i][1] * g.conv_stride + kh - g.conv_pad, xx = rowinfo[i][2] * g.conv_stride + kw - g.conv_pad;
                const int lim_h = g.conv_up ? 2 * g.conv_hin : g.conv_hin, lim_w = g.conv_up ? 2 * g.conv_win : g.conv_win;
                if (yy >= 0 && yy < lim_h && xx >= 0 && xx < lim_w) {
                    if (g.conv_up) { yy >>= 1; xx >>= 1; }
                    v = *reinterpret_cast<const float4*>(A + (((long)rowinfo[i][0] * g.conv_hin + yy) * g.conv_win + xx) * g.conv_cin + c0 + c4 * 4);
                }
            }
        }
        r.v[i] = v;
    }
}

__device__ __forceinline__ void load_b_tile(const GemmArgs& g, const float* __restrict__ B, int n0, int k0, int tid, Frag4& r) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = tid + 256 * i;
        const int row = e >> 3, c4 = e & 7;
        const int n = n0 + row;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n < g.N) v = *reinterpret_cast<const float4*>(B + (long)n * g.ldb + k0 + c4 * 4);
        r.v[i] = v;
    }
}

__device__ __forceinline__ void store_tile(float* __restrict__ S, int tid, const Frag4& r) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = tid + 256 * i;
        const int row = e >> 3, c4 = e & 7;
        *reinterpret_cast<float4*>(S + row * LDSS + c4 * 4) = r.v[i];
    }
}

template <int MODE>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                    // [2][BM][LDSS]
    float* Bs = smem + 2 * BM * LDSS;    // [2][BN][LDSS]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, h = lane >> 5;
    // N-tiles fastest within an XCD-contiguous id range: the column tiles of one A row-panel run back to back on ONE XCD (shared L2)
    int tx, ty;
    xcd_tile(gridDim.x, gridDim.y, tx, ty);
    const int n0 = tx * BN, m0 = ty * BM;
    const int bz = blockIdx.z;
    const float* A = g.A + (long)bz * g.strideA;
    const float* B = g.B + (long)bz * g.strideB;

    int rowinfo[4][3] = {};
    if (MODE == MODE_CONV3) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + ((tid + 256 * i) >> 3);
            const int hw = g.conv_h * g.conv_w;
            const int img = m / hw;
            const int rem = m - img * hw;
            rowinfo[i][0] = img;
            rowinfo[i][1] = rem / g.conv_w;
            rowinfo[i][2] = rem - rowinfo[i][1] * g.conv_w;
        }
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

    const int nk = g.K / BK;
    Frag4 ra, rb;
    load_a_tile<MODE>(g, A, m0, 0, tid, ra, rowinfo);
    load_b_tile(g, B, n0, 0, tid, rb);
    store_tile(As, tid, ra);
    store_tile(Bs, tid, rb);
    __syncthreads();

    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        if (more) {
            load_a_tile<MODE>(g, A, m0, (kt + 1) * BK, tid, ra, rowinfo);
            load_b_tile(g, B, n0, (kt + 1) * BK, tid, rb);
        }
        const float* as = As + cur * BM * LDSS + (wm * 64 + r) * LDSS + h * 4;
        const float* bs = Bs + cur * BN * LDSS + (wn * 64 + r) * LDSS + h * 4;
#pragma unroll
        for (int c = 0; c < BK / 8; ++c) {
            float4 a0 = *reinterpret_cast<const float4*>(as + c * 8);
            float4 a1 = *reinterpret_cast<const float4*>(as + 32 * LDSS + c * 8);
            float4 b0 = *reinterpret_cast<const float4*>(bs + c * 8);
            float4 b1 = *reinterpret_cast<const float4*>(bs + 32 * LDSS + c * 8);
            const float av0[4] = {a0.x, a0.y, a0.z, a0.w}, av1[4] = {a1.x, a1.y, a1.z, a1.w};
            const float bv0[4] = {b0.x, b0.y, b0.z, b0.w}, bv1[4] = {b1.x, b1.y, b1.z, b1.w};
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[kk], bv0[kk], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[kk], bv1[kk], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[kk], bv0[kk], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[kk], bv1[kk], acc[1][1], 0, 0, 0);
            }
        }
        if (more) {
            store_tile(As + (cur ^ 1) * BM * LDSS, tid, ra);
            store_tile(Bs + (cur ^ 1) * BN * LDSS, tid, rb);
        }
        __syncthreads();
        cur ^= 1;
    }

    // epilogue.  C/D layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
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
                float v = acc[i][j][q] * g.alpha + bn;
                if (g.bias_m) v += g.bias_m[m];
                if (g.act == ACT_GELU) v = gelu_erf(v);
                if (R) v += R[(long)m * g.ldr + n];
                C[(long)m * g.ldc + n] = v;
            }
        }
}

static GemmArgs conv_defaults(const GemmArgs& g0) {
    GemmArgs g = g0;
    if (g.mode == MODE_CONV3) {
        if (g.conv_stride == 0) g.conv_stride = 1;
        if (g.conv_pad < 0) g.conv_pad = 1;
        if (g.conv_hin == 0) g.conv_hin = g.conv_up ? g.conv_h / 2 : g.conv_h;
        if (g.conv_win == 0) g.conv_win = g.conv_up ? g.conv_w / 2 : g.conv_w;
    }
    return g;
}

// per calling thread (like the profiler hook): set by every entry point for the context it runs, so two contexts driven from two host threads do not see each other's table
static thread_local const std::unordered_map<const float*, SplitPlanes>* g_split_table = nullptr;
void split_registry_set(const void* table) { g_split_table = reinterpret_cast<const std::unordered_map<const float*, SplitPlanes>*>(table); }

void launch_gemm(const GemmArgs& g_in, hipStream_t stream) {
    const GemmArgs g = conv_defaults(g_in);
    if (g.B_hi) return g.A_hi ? launch_gemm_split_glds(g, stream) : launch_gemm_split(g, stream);
    if (g_split_table && g.strideB == 0) {  // the executing context runs in split-precision mode and B is one of its (pre-split) weights
        auto it = g_split_table->find(g.B);
        if (it != g_split_table->end()) {
            GemmArgs s = g;
            s.B_hi = it->second.hi;
            s.B_lo = it->second.lo;
            s.b_lo_zero = it->second.lo_zero;
            if (s.A_hi) return launch_gemm_split_glds(s, stream);
            return launch_gemm_split(s, stream);
        }
    }
    BG_REQUIRE(g.A != nullptr, "gemm: A was given as split planes but B has no split planes (weight not registered for the split-precision path)");
    BG_REQUIRE(g.K % BK == 0, "gemm: K=%d must be a multiple of %d (pad the operands)", g.K, BK);
    BG_REQUIRE(g.lda % 4 == 0 && g.ldb % 4 == 0, "gemm: lda/ldb must be multiples of 4 floats");
    BG_REQUIRE(g.M > 0 && g.N > 0 && g.batch > 0, "gemm: empty problem");
    if (g.mode == MODE_CONV3) BG_REQUIRE(g.conv_cin % BK == 0 && g.K == 9 * g.conv_cin, "conv3x3: Cin=%d must be a multiple of 32", g.conv_cin);
    dim3 grid(cdiv(g.N, BN), cdiv(g.M, BM), g.batch);
    const size_t lds = (size_t)2 * (BM + BN) * LDSS * sizeof(float);
    static std::atomic<bool> attr_set[kMaxDevices];
    const int dslot = device_slot();
    if (!attr_set[dslot].load(std::memory_order_acquire)) {
        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f32_kernel<MODE_PLAIN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f32_kernel<MODE_CONV3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set[dslot].store(true, std::memory_order_release);
    }
    ProfScope prof(g.mode == MODE_CONV3 ? PROF_CONV3 : PROF_GEMM, 2.0 * g.M * (double)g.N * g.K * g.batch, stream);
    if (g.mode == MODE_CONV3)
        hipLaunchKernelGGL(gemm_f32_kernel<MODE_CONV3>, grid, dim3(256), lds, stream, g);
    else
        hipLaunchKernelGGL(gemm_f32_kernel<MODE_PLAIN>, grid, dim3(256), lds, stream, g);
    LAUNCH_CHECK();
}

}  // namespace bevgen
