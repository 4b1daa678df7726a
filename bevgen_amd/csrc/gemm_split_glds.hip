// Split-precision GEMM, LDS-DMA variant: BOTH operands arrive as (hi, lo) f16 planes in HBM and are moved global -> LDS by
// global_load_lds_dwordx4 (no VGPR staging, no ds_write: on gfx950 a ds_write_b128 costs 13 cycles of the SIMD->LDS path and the
// register-staged kernel in gemm_split.hip spends more LDS-pipe time writing tiles than the matrix pipe spends on the MFMAs).
//
//   C[M,N] = (Ah + Al 2^-11)(Bh + Bl 2^-11)^T  ~=  Ah Bh^T + 2^-11 (Ah Bl^T + Al Bh^T)          (see gemm_split.hip for the numerics)
//
// * Block tile (WM*64) x 128 x 32: WM*2 waves, each a 64x64 patch = (2x2) v_mfma_f32_32x32x16_f16 tiles with a main and a correction
//   accumulator.  WM = 4 (256x128, 8 waves = two per SIMD, one block per CU) is the production shape.
// * Operands live in HBM in the interleaved plane layout [row][k/32][hi 32 | lo 32] (gemm_split.hip): one (row, k-block) is one 128-byte
//   line, so every DMA request is a whole line (with separate hi / lo planes each k-tile touched only half of every line it fetched and
//   the L2->L1 fill traffic was twice the useful bytes - the measured limiter).
// * LDS image: per stage WM*64 A rows and 128 B rows of 128 bytes, UNPADDED because an LDS-DMA instruction writes base + lane*16 linearly
//   (1 KiB = 8 rows per wave-instruction).  Bank conflicts of the ds_read_b128 operand fetch are removed by an XOR swizzle of the 16-byte
//   chunk index (0-3 hi, 4-7 lo) with bits 1-3 of the row, applied on the SOURCE address of the DMA and again on the read (the destination
//   stays linear; MI355X guide rule 21); the 16-lane service groups of ds_read_b128 then cover all 64 banks exactly once.
// * Three-stage ring, ONE s_barrier per k-tile, software pipelined so that the matrix pipe never waits for the LDS:
//        top of k-tile t :  ds_read fragments F1 (k-step 1 of tile t)            | MFMAs on F0 (k-step 0, fetched during tile t-1)
//        middle          :  lgkmcnt(0); vmcnt(tile t+1 landed); s_barrier        -- every wave is done READING tile t, tile t+1 is complete
//        bottom          :  ds_read F0 of tile t+1; DMA of tile t+3 -> stage of t | MFMAs on F1, DMA pieces interleaved between MFMA groups
//   so every ds_read is issued a full 12-MFMA phase before its use and every DMA two k-tiles before its use.
// * MODE_CONV3: implicit-GEMM 3x3 convolution over NHWC planes; taps that fall into the zero padding are fetched from a zero page.
// * MODE_CONV3S: the same for the common case (stride 1, padding 1, no fused upsample: every 3x3 convolution of the VQGAN decoder but the four behind an upsample).  The address
//   of a tap is then the output pixel's own address plus a WAVE-UNIFORM displacement, and whether the tap lies inside the image is one bit of a 9-bit mask the lane forms once:
//   4 VALU per DMA piece instead of ~20 (coordinates, four bound compares, the address polynomial), no per-piece coordinate registers - the general variant sits at the
//   register limit and spills 13 VGPRs, and the VALU in front of every DMA request delays the MFMAs queued behind it.
#include "common.h"
#include "kernels.h"
#include "profiler.h"
#include <type_traits>
#include <vector>
#include <cstdio>

namespace bevgen {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// sum over an aligned row of 16 lanes with DPP row operations (every lane of the row ends with the row's sum)
__device__ __forceinline__ float row16_sum(float d) {
    d += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(d), 0xB1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
    d += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(d), 0x4E, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
    d += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(d), 0x141, 0xf, 0xf, true));   // row_half_mirror
    d += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(d), 0x140, 0xf, 0xf, true));   // row_mirror
    return d;
}

// x / d for many x and ONE d (a row's l2norm): the compiler's IEEE division is ten instructions per quotient (v_div_scale x 2, v_rcp, four fma, v_div_fmas, v_div_fixup)
// and the fused q / k epilogues do 128 of them per lane - 5 of the 8 us that epilogue held the CU's VALUs with every matrix pipe idle.  The scaling and fix-up steps only
// serve operands near the ends of the exponent range; here d is in [1e-12, 1e6] and |x| below 1e6, so Markstein's sequence with a refined reciprocal gives the correctly
// rounded quotient in three instructions: q0 = x r, e = x - d q0 (exact in fma), q = q0 + e r.
struct RowDiv {
    float d, r;
    __device__ __forceinline__ explicit RowDiv(float d_) : d(d_) {
        float r0 = __builtin_amdgcn_rcpf(d_);
        const float e = fmaf(-d_, r0, 1.0f);
        r0 = fmaf(e, r0, r0);
        const float e2 = fmaf(-d_, r0, 1.0f);
        r = fmaf(e2, r0, r0);
    }
    __device__ __forceinline__ float operator()(float x) const {
        const float q0 = x * r;
        const float e = fmaf(-d, q0, x);
        return fmaf(e, r, q0);
    }
};
constexpr int GBN = 128, GBK = 32;
#ifdef BEVGEN_GEMM_TRACE   // tools/gemm_trace: phase stamps (100 MHz clock) of the first 2048 workgroups of the LAST throughput launch: entry, first tile landed, loop done, epilogue issued, stores acknowledged
__device__ unsigned long long g_gemm_trace[5 * 2048 * 8];   // [epilogue kind][workgroup][stamp]
__device__ unsigned long long g_gemm_tr_tmp[2];
#define GT_STAMP(i) do { if (MODE == MODE_PLAIN && !KS) gt[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define GT_STAMP(i) do { } while (0)
#endif
constexpr float kGLoInv = 1.f / 2048.f;

__device__ __forceinline__ void glds16(const _Float16* src, _Float16* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
// buffer form: wave-uniform resource + 32-bit per-lane byte offset (constant over k) + scalar byte offset (the k position)
__device__ __forceinline__ void glds16_buf(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, int soff, _Float16* lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, (int)voff, soff, 0, 0);
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// at most n (runtime, capped at MAXN) k-tiles of DMA wave-instructions each may stay in flight
template <int DMA, int MAXN>
__device__ __forceinline__ void wait_tiles(int n) {
    if constexpr (MAXN <= 0) wait_vmcnt<0>();
    else {
        if (n >= MAXN) wait_vmcnt<MAXN * DMA>();
        else wait_tiles<DMA, MAXN - 1>(n);
    }
}

// the same with X younger wave-instructions (the residual prefetch of the tail) allowed to stay in flight on top
template <int DMA, int MAXN, int X>
__device__ __forceinline__ void wait_tiles_x(int n) {
    if constexpr (MAXN <= 0) wait_vmcnt<X>();
    else {
        if (n >= MAXN) wait_vmcnt<MAXN * DMA + X>();
        else wait_tiles_x<DMA, MAXN - 1, X>(n);
    }
}

// W16: the B operand (a weight matrix) is f16-representable, its low plane is zero: the product is  b_hi a_hi + 2^-11 b_hi a_lo  - TWO MFMAs per
// k-step pair instead of three, and no low-plane fragment reads (weight_dtype = f16 contexts)
// KS: the k range is cut into gridDim.z slices (small-M problems; a template flag so that the throughput instantiations carry none of it - the 256-row convolution
// variant sits at the register limit and spilled 300 VGPRs with the slice arithmetic compiled in)
// TI: 32-row MFMA tiles per wave along M.  2 = the 64x64 wave patch.  1 = a 32x64 patch, twice the waves on the same block tile: the shape of a block that has its CU
// to itself (one or two scenes: fewer blocks than CUs).  With four waves every SIMD holds ONE wave, and the ~70 issue cycles of each of its 8 DMA instructions per k-tile
// are cycles in which that SIMD's matrix pipe has nothing queued: 0.55 us per k-tile measured against 0.32 us of MFMAs (tools/gemm_small_probe.py).  Eight waves halve
// both the MFMAs and the DMA instructions of a wave and give every SIMD a second wave to run while one issues.
// TJ: 32-column MFMA tiles per wave along N (2 = the wave's 64 columns are one head / one x|gate pair, which the fused epilogues rely on; 1 = 32x32 patches, plain
// epilogue only: the 64-row block of a lone small problem on eight waves instead of four)
// The body of one block tile: tile (tx, ty), k-tiles [kt_first, kt_first + nk).  ksl / kz: the split-K slice geometry of the KS instantiations (1 / 0 otherwise).
// sk_role (stream-K launches, gemm_split_glds_sk_kernel below): 0 = the whole k range of the tile is here: the normal epilogue; 1 = a PART of the tile's k range whose last
// part lies with a later workgroup: the raw tile sums go to this workgroup's slot of g.sk_ws and its flag is raised; 2 = the LAST part: the sums of the workgroups
// sk_first .. blockIdx.x - 1 (ascending k) are added in that order, then the normal epilogue.
template <int MODE, int WM, int S, bool W16, bool KS, int TI, int TJ, bool SKK = false>
__device__ __forceinline__ void gemm_tile(const GemmArgs& g, const int tx, const int ty, const int kt_first, const int nk, const int ksl, const int kz, const int sk_role,
                                          const int sk_first, const int tid
#ifdef BEVGEN_GEMM_TRACE
                                          , unsigned long long* gt
#endif
                                          ) {
    constexpr int TBM = WM * 64;                 // block rows
    constexpr int NW = WM * 8 / (TI * TJ);       // waves
    constexpr int WROWS = TI * 32;               // rows of a wave's patch
    constexpr int WCOLS = TJ * 32;               // columns of a wave's patch
    constexpr int WN = GBN / WCOLS;              // waves along N
    constexpr int NAJ = TBM / 8 / NW;            // 8-row A pieces per wave
    constexpr int NBJ = 16 / NW;                 // 8-row B pieces per wave
    constexpr int AREGION = TBM * 2 * GBK;       // halves: TBM rows x (32 hi | 32 lo)
    constexpr int STAGE_H = AREGION + GBN * 2 * GBK;  // halves per stage
    constexpr int DMA = NAJ + NBJ;                    // DMA wave-instructions per k-tile
    extern __shared__ __attribute__((aligned(1024))) _Float16 smem_g[];
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int r = lane & 31, h = lane >> 5;
    const int n0 = tx * GBN, m0 = ty * TBM + g.m_base;
    const int n0l = n0, m0l = m0;

    const _Float16* Ap = reinterpret_cast<const _Float16*>(g.A_hi);   // interleaved planes: (row, 32-k block) = 64 halves = one 128-byte line
    const _Float16* Bp = reinterpret_cast<const _Float16*>(g.B_hi);

    // ---- DMA descriptors of this lane.  One DMA piece = 8 rows x 128 B (hi|lo of one k-block): lane -> row lane>>3, 16-byte position lane&7.
    // The wave owns TBM/NW A rows (NAJ pieces) and 128/NW B rows (NBJ pieces).
    // MODE_PLAIN: buffer_load ... lds with address = resource base + scalar k offset + a per-lane 32-bit byte offset that never changes:
    // no per-iteration address VALU and 6 VGPRs of DMA state.  Rows outside the problem are CLAMPED to the last row: they only feed
    // accumulator rows / columns that the epilogue never stores.
    // MODE_CONV3: the A address depends on the tap (and taps inside the zero padding read a zero page), so it is rebuilt per piece.
    constexpr bool CONVG = MODE == MODE_CONV3, CONVS = MODE == MODE_CONV3S, CONV = CONVG || CONVS;
    unsigned a_off[NAJ], b_off[NBJ];
    int a_img[CONVG ? NAJ : 1], a_yx[CONVG ? NAJ : 1];   // MODE_CONV3: image index and (y << 16 | x) of the output pixel; rows past M have a_img >= number of images
    unsigned a_msk[CONVS ? NAJ : 1];                      // MODE_CONV3S: bit (3 kh + kw) = tap (kh, kw) of this lane's output pixel lies inside the image (0 for rows past M)
#pragma unroll
    for (int j = 0; j < NBJ; ++j) {
        const int R = wave * (8 * NBJ) + j * 8 + (lane >> 3);   // B row inside the tile
        const int c = (lane & 7) ^ ((R >> 1) & 7);               // logical 16-byte chunk (0-3 hi, 4-7 lo) that lives at physical position lane&7 of row R
        const int n = min(n0l + R, g.N - 1);
        b_off[j] = (unsigned)(((long)n * 2 * g.ldb + c * 8) * 2);
    }
#pragma unroll
    for (int j = 0; j < NAJ; ++j) {
        const int R = wave * (8 * NAJ) + j * 8 + (lane >> 3);   // A row inside the tile
        const int c = (lane & 7) ^ ((R >> 1) & 7);
        const int m = m0l + R;
        if constexpr (MODE == MODE_PLAIN) {
            a_off[j] = (unsigned)(((long)min(m, g.M - 1) * 2 * g.lda + c * 8) * 2);
        } else if constexpr (CONVS) {
            const int hw = g.conv_h * g.conv_w;
            const int mc = min(m, g.M - 1);
            const int img = mc / hw, rem = mc - img * hw;
            const int y = rem / g.conv_w, x = rem - y * g.conv_w;
            a_off[j] = (unsigned)mc * (unsigned)(g.conv_cin * 4) + (unsigned)(c * 16);   // the pixel's own (centre tap) line, this lane's 16-byte chunk
            unsigned rowm = (y > 0 ? 1u : 0u) | 2u | (y + 1 < g.conv_h ? 4u : 0u);       // kh = 0, 1, 2 inside
            unsigned colm = (x > 0 ? 1u : 0u) | 2u | (x + 1 < g.conv_w ? 4u : 0u);       // kw = 0, 1, 2 inside
            unsigned msk = 0;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
                if (rowm & (1u << kh)) msk |= colm << (3 * kh);
            a_msk[j] = m < g.M ? msk : 0u;
        } else {
            a_off[j] = (unsigned)(c * 16);   // byte position of the lane's chunk inside the 128-byte (pixel, k-block) line
            const int hw = g.conv_h * g.conv_w;
            const int img = m / hw, rem = m - img * hw;
            const int y = rem / g.conv_w;
            a_img[j] = img; a_yx[j] = (y << 16) | (rem - y * g.conv_w);
        }
    }
    // DMA of the NEXT k-tile, in pieces (tiles are issued strictly in order)
    int k_issue = kt_first * GBK;
    const int n_img = CONVG ? g.M / (g.conv_h * g.conv_w) : 0;
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(Ap), 0, CONV ? g.a_bytes : -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(Bp), 0, -1, 0x00020000);
    // MODE_CONV3S: wave-uniform state of the k-tile being issued - its tap as a mask bit, and the byte displacement of (tap, channel block) from the centre line
    int s_c0 = 0, s_tap = 0;
    if (CONVS && kt_first > 0) { s_tap = k_issue / g.conv_cin; s_c0 = k_issue - s_tap * g.conv_cin; }
    auto tap_delta = [&]() { const int kh = s_tap / 3, kw = s_tap - 3 * kh; return ((kh - 1) * g.conv_w + (kw - 1)) * (g.conv_cin * 4) + s_c0 * 4; };
    int s_delta = CONVS ? tap_delta() : 0;
    unsigned s_bit = 1u << s_tap;
    auto issue_a = [&](int stage, int j) {
        _Float16* sa = smem_g + stage * STAGE_H + (wave * (8 * NAJ) + j * 8) * 2 * GBK;
        if constexpr (CONVS) {
            // (an offset beyond the resource's num_records returns zeros, as in the general variant)
            glds16_buf(a_rsrc, (a_msk[j] & s_bit) ? a_off[j] + (unsigned)s_delta : 0xFFFFFF00u, 0, sa);
        } else if constexpr (CONVG) {
            // scalar part (tap of this k-tile) + a dozen 32-bit VALU per piece; taps inside the zero padding (and rows past M) use an offset
            // beyond the resource's num_records: the buffer load returns zeros and the DMA writes them (checked on gfx950)
            const int tap = k_issue / g.conv_cin;
            const int c0 = k_issue - tap * g.conv_cin;
            const int kh = tap / 3, kw = tap - kh * 3;
            int yy = (a_yx[j] >> 16) * g.conv_stride + kh - g.conv_pad, xx = (a_yx[j] & 0xffff) * g.conv_stride + kw - g.conv_pad;
            const int lim_h = g.conv_up ? 2 * g.conv_hin : g.conv_hin, lim_w = g.conv_up ? 2 * g.conv_win : g.conv_win;
            const bool ok = a_img[j] < n_img && yy >= 0 && yy < lim_h && xx >= 0 && xx < lim_w;
            if (g.conv_up) { yy >>= 1; xx >>= 1; }
            const unsigned off = (unsigned)((a_img[j] * g.conv_hin + yy) * g.conv_win + xx) * (unsigned)(g.conv_cin * 4) + (unsigned)(c0 * 4) + a_off[j];
            glds16_buf(a_rsrc, ok ? off : 0xFFFFFF00u, 0, sa);
        } else {
            glds16_buf(a_rsrc, a_off[j], k_issue * 4, sa);   // (k/32) * 128 bytes
        }
    };
    auto issue_b = [&](int stage) {   // last piece(s) of a tile: advances the k position
        _Float16* sb = smem_g + stage * STAGE_H + AREGION + wave * (8 * NBJ) * 2 * GBK;
#pragma unroll
        for (int j = 0; j < NBJ; ++j) glds16_buf(b_rsrc, b_off[j], k_issue * 4, sb + j * 8 * 2 * GBK);
        k_issue += GBK;
        if (CONVS) {
            s_c0 += GBK;
            s_delta += GBK * 4;
            if (s_c0 == g.conv_cin) { s_c0 = 0; ++s_tap; s_bit <<= 1; s_delta = tap_delta(); }
        }
    };
    auto issue_a_half = [&](int stage, int half) {   // the A pieces in two groups (interleaved with the MFMA groups of a phase)
#pragma unroll
        for (int j = half * NAJ / 2; j < (half + 1) * NAJ / 2; ++j) issue_a(stage, j);   // (NAJ = 1: the piece goes with the second group)
    };

    f32x16 accM[TI][TJ], accC[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) { accM[i][j][q] = 0.f; accC[i][j][q] = 0.f; }

    // operand fetch addresses (halves) inside a stage: row*64 + swizzled chunk*8; the lo chunk (logical +4) sits at the hi address ^ 32 halves
    int a_rd[2][2], b_rd[2][2];   // [tile i/j][k-step]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int ra = wm * WROWS + (i < TI ? i : 0) * 32 + r, rb = wn * WCOLS + (i < TJ ? i : 0) * 32 + r;
            a_rd[i][ks] = ra * 2 * GBK + (((ks * 2 + h) ^ ((ra >> 1) & 7)) << 3);
            b_rd[i][ks] = AREGION + rb * 2 * GBK + (((ks * 2 + h) ^ ((rb >> 1) & 7)) << 3);
        }
    struct Frag { half8 ah[2], al[2], bh[2], bl[2]; };
    auto fetch = [&](Frag& f, int stage, int ks) {
        const _Float16* st = smem_g + stage * STAGE_H;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (i < TI) {
                f.ah[i] = *reinterpret_cast<const half8*>(st + a_rd[i][ks]);
                f.al[i] = *reinterpret_cast<const half8*>(st + (a_rd[i][ks] ^ 32));
            }
            if (i < TJ) {
                f.bh[i] = *reinterpret_cast<const half8*>(st + b_rd[i][ks]);
                if (!W16) f.bl[i] = *reinterpret_cast<const half8*>(st + (b_rd[i][ks] ^ 32));
            }
        }
    };
    auto mma_main = [&](const Frag& f) {
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j) accM[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.bh[j], f.ah[i], accM[i][j], 0, 0, 0);
    };
    auto mma_c1 = [&](const Frag& f) {
        if (W16) return;
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j) accC[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.bl[j], f.ah[i], accC[i][j], 0, 0, 0);
    };
    auto mma_c2 = [&](const Frag& f) {
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j) accC[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.bh[j], f.al[i], accC[i][j], 0, 0, 0);
    };

    // ---- prologue: tiles 0..S-1 in flight, tile 0 landed, F0 of tile 0 on its way
#pragma unroll
    for (int s = 0; s < S; ++s)
        if (s < nk) { issue_a_half(s, 0); issue_a_half(s, 1); issue_b(s); }
    // Folded LayerNorm, consumer side (GemmArgs::ln_in_*): (mean, rstd) of this wave's rows, parked in LDS behind the stage ring (the wave's own 64 slots: the throughput
    // instantiation has no VGPR to carry them through its main loop) - requested HERE, in the shadow of the first tiles' DMA latency.  Either launch_ln_stats_finalize has
    // merged the producer's group sums (ln_in_stats: large batches - one small kernel per LayerNorm instead of a merge per tile), or this wave merges them itself
    // (ln_in_gsums: one or two scenes, where every workgroup runs one tile and a launch costs more than the merge): lane l takes row l of the wave's 64 (32-row patches: the
    // lane halves split the groups by parity), sixteen loads in flight, fp64 sums in a fixed order.
    float* ln_rowstat = reinterpret_cast<float*>(smem_g + S * STAGE_H);   // [TBM rows][mean, rstd] of the block's rows
    if (MODE == MODE_PLAIN && g.ln_in_stats) {
        if (tid < TBM) {
            const int m = min(m0 + tid, g.M - 1);
            *reinterpret_cast<float2*>(ln_rowstat + tid * 2) = reinterpret_cast<const float2*>(g.ln_in_stats)[m];
        }
    } else if (MODE == MODE_PLAIN && g.ln_in_gsums) {
        // the whole block merges together: thread -> (row tid % TBM, part tid / TBM); a part takes every PARTS-th group (one or two batches of loads in flight, where
        // a wave merging its own rows needed three to six round trips: 5.3 -> ~2 us of ring-fill time at one scene's down-projection); fp64 partial sums meet in LDS
        // (8 KB behind the LayerNorm slots and the prefetch sink) and are added in part order - a fixed order
        constexpr int NT = NW * 64, PARTS = NT / TBM;
        static_assert(NT % TBM == 0, "threads per block row");
        const int row = tid % TBM, part = tid / TBM;
        const int m = min(m0 + row, g.M - 1);
        const float2* st = reinterpret_cast<const float2*>(g.ln_in_gsums) + m;
        const int G = g.ln_in_groups;
        double s1 = 0.0, s2 = 0.0;
        for (int g0 = part; g0 < G; g0 += 16 * PARTS) {
            float2 v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = st[(long)min(g0 + u * PARTS, G - 1) * g.ln_rows];   // (clamped, never predicated: sixteen independent loads)
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (g0 + u * PARTS < G) { s1 += (double)v[u].x; s2 += (double)v[u].y; }
        }
        double* partial = reinterpret_cast<double*>(reinterpret_cast<char*>(smem_g + S * STAGE_H) + 6144);   // [PARTS][TBM][2]
        partial[(part * TBM + row) * 2] = s1;
        partial[(part * TBM + row) * 2 + 1] = s2;
        __syncthreads();
        if (tid < TBM) {
            double t1 = 0.0, t2 = 0.0;
#pragma unroll
            for (int pp = 0; pp < PARTS; ++pp) { t1 += partial[(pp * TBM + tid) * 2]; t2 += partial[(pp * TBM + tid) * 2 + 1]; }
            const double mean = t1 / (double)g.ln_in_count;
            const double var = fmax(t2 / (double)g.ln_in_count - mean * mean, 0.0);
            *reinterpret_cast<float2*>(ln_rowstat + tid * 2) = make_float2((float)mean, (float)(1.0 / sqrt(var + (double)g.ln_eps)));
        }
        // (the k loop's barriers order these writes before the epilogue's reads)
    }
    wait_tiles<DMA, S - 1>(nk - 1);   // tile 0 landed; the other tiles of the prologue stay in flight
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    GT_STAMP(1);
    Frag f0, f1;
    fetch(f0, 0, 0);

    // One k-tile.  WHERE selects the DMA work of this iteration at compile time (the steady-state loops below are branch free; a runtime
    // `if (more)` around every DMA piece costs ~10 %: each one ends a scheduling region in the middle of the MFMA stream):
    //   0  none (tail)      1  tile kt+S after the barrier, interleaved with the F1 MFMAs ("early" waves)
    //   2  tile kt+S-1 before the barrier, interleaved with the F0 MFMAs ("late" waves: the same tile, one phase later)
    // DMA issue is staggered because a glds costs its wave ~70 issue cycles during which the MFMAs queued behind it cannot start: of the
    // two waves sharing a SIMD one always runs bare MFMAs.   STEADY: tile kt+2 exists, so one tile may stay in flight across the barrier.
    // Residual prefetch (GemmArgs::r_prefetch).  The epilogue's residual rows (x = x + proj) are the one operand nobody asked for before the k loop ends: 256 workgroups
    // finish a round of tiles together and then all wait for 32 MB of residual lines at once with every matrix pipe idle.  In the FIRST iteration of the loop's tail
    // (no operand DMA left to issue, S - 1 .. S k-tiles before the epilogue) every wave requests the 128 lines of its 64 x 64 patch - two LDS-DMA instructions of one
    // 16-byte piece per line into a 2 KiB sink behind the ring; the data is never read from there, the lines are then in the XCD's L2 / the memory-side
    // cache when the epilogue loads them.  The tail's waits let these RPF_N youngest instructions stay in flight.
    constexpr bool RPF = MODE == MODE_PLAIN && TI == 2 && TJ == 2 && !KS && !SKK;   // (SKK: the call comes from the stream-K kernel)
    constexpr int RPF_N = 2;
    const bool rpf_on = RPF && g.r_prefetch && sk_role == 0 && nk >= S;
    bool rpf_pending = false, rpf_flight = false;
    auto issue_rpf = [&](int stg) {
        const __amdgpu_buffer_rsrc_t r_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.R), 0, -1, 0x00020000);
        const int row = min(m0 + wm * WROWS + lane, g.M - 1);
        const int col = min(n0 + wn * WCOLS, g.N - 64);
        const unsigned off = (unsigned)(((long)row * g.ldr + col) * 4);
        // (the sink: 2 KiB behind the ring and the LayerNorm slots, shared by all waves - nobody reads it.  NOT a free ring stage: a slow wave's pieces could land
        // there after a fast wave has begun to stage its epilogue tile in the ring)
        _Float16* dst = smem_g + S * STAGE_H + 2048;
        (void)stg;
        glds16_buf(r_rsrc, off, 0, dst);
        glds16_buf(r_rsrc, off + 128u, 0, dst + 512);
    };
    // (Measured and removed, round 6: a NEXT-TILE prefetch on the same pattern - in the last tail iteration every wave requested the lines of the first k-tiles of the tile
    // that the workgroup 256 places later runs on this XCD.  [24576, 1024] x 1024 153 -> 154 us, x 3072 475 -> 489 us, sixteen scenes 10.63 -> 10.56 scenes/s: the ring
    // fill of a round is not what its first microseconds wait for; profiles/r06_ab_gemm_npf.txt.)
    int stage = 0;
    auto body = [&](auto where_c, auto steady_c, int kt) {
        constexpr int WHERE = decltype(where_c)::value;
        constexpr bool STEADY = decltype(steady_c)::value;
        const int stage_next = stage == S - 1 ? 0 : stage + 1;
        const int stage_prev = stage == 0 ? S - 1 : stage - 1;
#define BG_FENCE() __builtin_amdgcn_sched_barrier(0)   /* pin the hand-placed order: the machine scheduler otherwise sinks the ds_reads below the MFMAs */
        fetch(f1, stage, 1);
        BG_FENCE();
        mma_main(f0);
        BG_FENCE();
        if (WHERE == 2) { issue_a_half(stage_prev, 0); BG_FENCE(); }
        mma_c1(f0);
        BG_FENCE();
        if (WHERE == 2) { issue_a_half(stage_prev, 1); BG_FENCE(); }
        mma_c2(f0);
        if (WHERE == 2) { BG_FENCE(); issue_b(stage_prev); }
        // this wave is done reading tile kt; tile kt+1 must be complete (own pieces; tiles kt+2 .. kt+S-1 may stay in flight).  The scheduling fences keep
        // the MFMAs of F0 above the waits (they are not memory operations, nothing else would stop them sinking below) and F1's below.
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (STEADY) wait_vmcnt<(S - 2) * DMA>();
        else if (RPF && rpf_flight) wait_tiles_x<DMA, S - 2, RPF_N>(nk - kt - 2);
        else wait_tiles<DMA, S - 2>(nk - kt - 2);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        mma_main(f1);                    // F1 landed before the barrier: the matrix pipe restarts at once
        __builtin_amdgcn_sched_barrier(0);
        fetch(f0, stage_next, 0);        // unconditional (after the last tile it reads a stale stage and is never used): keeps the wait counters exact
        BG_FENCE();
        if (WHERE == 1) { issue_a_half(stage, 0); BG_FENCE(); }
        if (RPF && WHERE == 0 && rpf_pending) { issue_rpf(stage); rpf_pending = false; rpf_flight = true; BG_FENCE(); }
        mma_c1(f1);
        BG_FENCE();
        if (WHERE == 1) { issue_a_half(stage, 1); BG_FENCE(); }
        mma_c2(f1);
        BG_FENCE();
        if (WHERE == 1) { issue_b(stage); BG_FENCE(); }
#undef BG_FENCE
        stage = stage_next;
    };
    using std::integral_constant;
    const bool late = NW == 8 && wave >= NW / 2;   // staggering only matters when two waves of ONE block share a SIMD
    int kt = 0;
    if (!late) {
        for (; kt < nk - S; ++kt) body(integral_constant<int, 1>{}, integral_constant<bool, true>{}, kt);   // tiles kt+S <= nk-1 exist
    } else {
        if (nk > 0) { body(integral_constant<int, 0>{}, integral_constant<bool, false>{}, 0); kt = 1; }
        for (; kt <= nk - S; ++kt) body(integral_constant<int, 2>{}, integral_constant<bool, true>{}, kt);  // tiles kt+S-1 <= nk-1 exist
    }
    rpf_pending = rpf_on;
    for (; kt < nk; ++kt) body(integral_constant<int, 0>{}, integral_constant<bool, false>{}, kt);
    GT_STAMP(2);

    // ---- stream-K (gemm_split_glds_sk_kernel): partial tile sums travel through g.sk_ws = [1024 flag words][workgroup slots of NW x TI x TJ x 16 x 64 floats].  The
    // workgroups that share a tile sit on ONE XCD (indices congruent mod 8: the kernel below deals whole tiles to XCDs), so the exchange stays in that XCD's L2 - plain
    // stores (the vector L1 is write-through: an acknowledged store is in L2), one flag per workgroup holding this launch's epoch (L2 atomic), sc1 loads of the slot (served
    // by the L2 whatever the reader's L1 holds): the protocol of ar_mlp_fused_kernel.  (First form of this kernel: consecutive workgroups = eight different XCDs, slots
    // written and read with sc1 at the memory side - 64 MB of exchange traffic per projection, one scene 162 -> 209 ms; profiles/r06_ab_gemm_sk_memside.txt.)
    // A workgroup only ever waits for LOWER-indexed ones, which were dispatched no later than itself and never wait for it: no deadlock even if the grid is not resident
    // at once.  The spin is bounded all the same (the status word reports a timeout).
    if (MODE == MODE_PLAIN && sk_role != 0) {
        constexpr int SLOT = NW * TI * TJ * 16 * 64;   // floats per workgroup
        float* ws = reinterpret_cast<float*>(g.sk_ws) + 1024;
        unsigned* flags = reinterpret_cast<unsigned*>(g.sk_ws);
        // slot layout: [wave][tile i][tile j][register quad qq][lane][4]: one 16-byte access per lane and quad, a wave's instruction covers 1 KiB
        const long lane_off = (long)wave * (TI * TJ * 16 * 64) + lane * 4;
        if (sk_role == 1) {
            float* mine = ws + (long)blockIdx.x * SLOT + lane_off;
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) {
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = accM[i][j][qq * 4 + e] + accC[i][j][qq * 4 + e] * kGLoInv;
                        *reinterpret_cast<f32x4*>(mine + ((i * TJ + j) * 4 + qq) * 256) = v;
                    }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (every wave: its stores are acknowledged)
            __syncthreads();
            if (tid == 0) __hip_atomic_store(flags + blockIdx.x, g.sk_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        // role 2: wait for the contributors (their partial was the FIRST thing they computed), add their sums in ascending k order, then this workgroup's own (the last k part)
        if (tid == 0) {
            const long long t_in = __builtin_amdgcn_s_memrealtime();
            for (int w = sk_first; w < (int)blockIdx.x; w += 8)
                while (__hip_atomic_load(flags + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != g.sk_epoch) {
                    __builtin_amdgcn_s_sleep(1);
                    if (__builtin_amdgcn_s_memrealtime() - t_in > 200LL * 1000 * 100) { status_raise(g.status, BG_ST_MLP_BARRIER); break; }   // 200 ms of the 100 MHz clock
                }
        }
        __syncthreads();
        // own sums first (main + correction accumulators merged: the correction registers are free from here on), then the contributors' added ONTO them from the nearest
        // k range down to the first - a fixed order; four register quads (4 KiB per wave) in flight at a time: the accumulators leave no room for more
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j)
#pragma unroll
                for (int q = 0; q < 16; ++q) { accM[i][j][q] += accC[i][j][q] * kGLoInv; accC[i][j][q] = 0.f; }
        for (int w = (int)blockIdx.x - 8; w >= sk_first; w -= 8) {
            const float* theirs = ws + (long)w * SLOT + lane_off;
            // half of a contributor's slot in flight at a time (8 KiB per wave: two L2 round trips per contributor; four quads at a time made it sixteen, the whole slot
            // does not fit beside the accumulators)
            constexpr int NQ = TI * TJ * 4, HB = NQ >= 8 ? 8 : NQ;
#pragma unroll
            for (int u0 = 0; u0 < NQ; u0 += HB) {
                f32x4 pv[HB];
#pragma unroll
                for (int u = 0; u < HB; ++u) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(pv[u]) : "v"(theirs + (u0 + u) * 256) : "memory");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                for (int u = 0; u < HB; ++u) {
                    asm volatile("" : "+v"(pv[u]));   // (defined behind the wait)
                    const int uu = u0 + u;            // = (i TJ + j) 4 + qq
#pragma unroll
                    for (int e = 0; e < 4; ++e) accM[uu / (TJ * 4)][(uu / 4) % TJ][(uu & 3) * 4 + e] += pv[u][e];
                }
            }
        }
    }
    // ---- folded LayerNorm, consumer side: LN(x) W^T = rstd (x (W o gamma)^T - mean cs) applied to the tile sums in place, so that every epilogue below sees the projection
    // of the normalised rows (it is linear in the sums: alpha, bias, activation and residual follow unchanged)
    if (MODE == MODE_PLAIN && (g.ln_in_stats || g.ln_in_gsums)) {
        float ln_mean[TI], ln_rstd[TI];
#pragma unroll
        for (int i = 0; i < TI; ++i) { const float2 mr = *reinterpret_cast<const float2*>(ln_rowstat + (wm * WROWS + i * 32 + r) * 2); ln_mean[i] = mr.x; ln_rstd[i] = mr.y; }
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j)
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    const int n = min(n0 + wn * WCOLS + j * 32 + 8 * qq + 4 * h, g.N - 4);   // (columns past N feed accumulator columns no epilogue stores)
                    const f32x4 cs = *reinterpret_cast<const f32x4*>(g.ln_in_cs + n);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int q = qq * 4 + e;
                        accM[i][j][q] = ln_rstd[i] * ((accM[i][j][q] + accC[i][j][q] * kGLoInv) - ln_mean[i] * cs[e]);
                        accC[i][j][q] = 0.f;
                    }
                }
    }
    // ---- epilogue.  The MFMAs were issued with the operands swapped (B fragment first), so the accumulators hold the TRANSPOSED 32x32
    // tile: lane -> output row m = lane&31, register q -> column n = (q&3) + 8*(q>>2) + 4*(lane>>5).  Four consecutive registers are four
    // consecutive columns of one row: one 16-byte store per lane and register quad (16 store instructions per wave instead of 64 - the
    // store tail is instruction-issue bound, MI355X guide T21).
    // Row-major store forms (STG): a wave stages its patch (64 x 64 in the throughput instantiation: 18 KB) in its own slice of the dead stage ring (row stride 68 floats) in the
    // accumulator layout and stores it from a row-major view - see the plain epilogue below for the measurement behind it.  The per-row arithmetic (l2norm, GEGLU, the
    // LayerNorm statistics) stays on the accumulator side, element for element as in the direct forms: results are bit-identical, only the store instructions change.
    // (Also the small-problem shapes - 32-row patches of the eight-wave / 64-row blocks, 32 x 32 patches with the plain epilogue - wherever the ring holds the slices; not
    // the two-stage 128-row block, whose 64 KB ring is smaller than its four slices.)
    constexpr int VT_LD = WROWS == 64 ? 72 : 40;          // transposed value stage: [64 columns][1 + WROWS tokens], padded (4 VT_LD = 32 mod 64 banks)
    constexpr int RS = TJ == 2 ? 68 : 36;                 // row stride (floats) of the row-major stage
    constexpr int SLICE_W = TJ == 2 ? (64 * VT_LD > WROWS * RS ? 64 * VT_LD : WROWS * RS) : WROWS * RS;   // words per wave (a tile may hold both kinds of waves: one size)
    constexpr bool STG = MODE == MODE_PLAIN && !KS && !SKK && !(WM == 2 && S == 2) && (TI == 2 ? TJ == 2 : true) && (long)NW * SLICE_W * 4 <= (long)S * STAGE_H * 2;
    float* const pl = reinterpret_cast<float*>(smem_g) + wave * SLICE_W;
    // the row-major view of a 64-column patch: 8 lanes x 8 columns per row, 8 rows per pass
    const int rm_row = lane >> 3, rm_c8 = lane & 7;
    auto stage_read8 = [&](int row, f32x4& a0, f32x4& a1) {
        a0 = *reinterpret_cast<const f32x4*>(pl + row * RS + 8 * rm_c8);
        a1 = *reinterpret_cast<const f32x4*>(pl + row * RS + 8 * rm_c8 + 4);
    };
    auto split8 = [&](const f32x4& a0, const f32x4& a1, half8& hi8, half8& lo8) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            hi8[e] = split_hi(a0[e]); lo8[e] = split_lo(a0[e], hi8[e]);
            hi8[4 + e] = split_hi(a1[e]); lo8[4 + e] = split_lo(a1[e], hi8[4 + e]);
        }
    };
    if constexpr (TJ == 2) {   // the fused epilogues and the split-K partial store assume 64-column wave patches
    // (a width that is not a multiple of 128 - dim 192: three heads - leaves the last tile's second wave without columns: its 64-column patch is not a head / an x|gate pair)
    if (MODE == MODE_PLAIN && g.epi != EPI_PLAIN && n0 + wn * 64 >= g.N) return;
    const bool qkv = MODE == MODE_PLAIN && g.epi == EPI_MUSE_QKV;
    if (g.epi == EPI_MUSE_Q || (qkv && n0 + wn * 64 < g.epi_heads * 64)) {
        // Route M query preparation fused into the to_q projection (muse_net:132-137; replaces muse_q_prep_split): the wave's 64 columns are
        // exactly one head, so q = l2norm(8 x) * q_scale is a per-lane reduction over its 32 registers plus one lane-half exchange; the
        // result leaves as the (hi, lo) f16 planes [B, H, Nq, 64] the attention kernel reads.
        _Float16* Qh = reinterpret_cast<_Float16*>(qkv ? g.epi_qh : g.epi_hi);
        _Float16* Ql = reinterpret_cast<_Float16*>(qkv ? g.epi_ql : g.epi_lo);
        const float* qsc = qkv ? g.epi_qscale : g.epi_scale;
        const int head = (n0 + wn * 64) >> 6;
        unsigned bad = 0;   // a prepared operand that is NaN / outside the f16 range (l2norm bounds q and k: only a non-finite projection gets here; v is unbounded)
        f32x4 qsc4[2][4];   // (the per-dimension scales, loaded in front of the first store: loads and stores share vmcnt - see the plain epilogue)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) qsc4[j][qq] = *reinterpret_cast<const f32x4*>(qsc + j * 32 + 8 * qq + 4 * h);
        if constexpr (STG) {
            if (g.row_major_epi) {
#pragma unroll
                for (int i = 0; i < TI; ++i) {
                    float v[2][16];
                    float ss = 0.f;
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int q = 0; q < 16; ++q) {
                            v[j][q] = (accM[i][j][q] + accC[i][j][q] * kGLoInv) * g.alpha * 8.0f;
                            ss = fmaf(v[j][q], v[j][q], ss);
                        }
                    ss += xor32(ss);
                    const float nrm = fmaxf(sqrtf(ss), 1e-12f);
                    const RowDiv rdiv(nrm);
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int qq = 0; qq < 4; ++qq) {
                            const f32x4 sc = qsc4[j][qq];
                            f32x4 o;
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[e] = (rdiv(v[j][qq * 4 + e]) * sc[e]) * g.epi_post;
                            *reinterpret_cast<f32x4*>(pl + (i * 32 + r) * RS + j * 32 + 8 * qq + 4 * h) = o;
                        }
                }
#pragma unroll
                for (int ps = 0; ps < WROWS / 8; ++ps) {
                    const int row = ps * 8 + rm_row, m = m0 + wm * WROWS + row;
                    f32x4 a0, a1;
                    stage_read8(row, a0, a1);
                    half8 hi8, lo8;
                    split8(a0, a1, hi8, lo8);
                    if (m < g.M) {
                        guard_half8(hi8, bad);
                        const int bb = m / g.epi_rows, nq = m - bb * g.epi_rows;
                        const long dst = (((long)bb * g.epi_heads + head) * g.epi_rows + nq) * 64 + 8 * rm_c8;
                        *reinterpret_cast<half8*>(Qh + dst) = hi8;
                        *reinterpret_cast<half8*>(Ql + dst) = lo8;
                    }
                }
                if (bad) status_raise(g.status, BG_ST_F16_RANGE);
                return;
            }
        }
#pragma unroll
        for (int i = 0; i < TI; ++i) {
            const int m = m0 + wm * WROWS + i * 32 + r;
            float v[2][16];
            float ss = 0.f;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    v[j][q] = (accM[i][j][q] + accC[i][j][q] * kGLoInv) * g.alpha * 8.0f;
                    ss = fmaf(v[j][q], v[j][q], ss);
                }
            ss += xor32(ss);
            const float nrm = fmaxf(sqrtf(ss), 1e-12f);
                    const RowDiv rdiv(nrm);
            if (m >= g.M) continue;
            const int bb = m / g.epi_rows, nq = m - bb * g.epi_rows;
            const long dst = (((long)bb * g.epi_heads + head) * g.epi_rows + nq) * 64;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    const int d = j * 32 + 8 * qq + 4 * h;
                    const f32x4 sc = qsc4[j][qq];
                    half4_t hi4, lo4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float qv = (rdiv(v[j][qq * 4 + e]) * sc[e]) * g.epi_post;   // epi_post: the attention's score scale, folded into q
                        hi4[e] = split_hi(qv);
                        lo4[e] = split_lo(qv, hi4[e]);
                    }
                    guard_half4(hi4, bad);
                    *reinterpret_cast<half4_t*>(Qh + dst + d) = hi4;
                    *reinterpret_cast<half4_t*>(Ql + dst + d) = lo4;
                }
        }
        if (bad) status_raise(g.status, BG_ST_F16_RANGE);
        return;
    }
    if (MODE == MODE_PLAIN && (g.epi == EPI_MUSE_KV || qkv)) {
        // Route M key / value preparation fused into the to_kv projection (muse_net:132-146; replaces muse_kv_prep_split and its pass over the raw projection):
        // the wave's 64 columns are one head of k (columns < H*64) or of v.  k: l2norm (eps 1e-12) * k_scale, hi/lo planes [B, H, ld, 64] at key row 1 + token.
        // v: hi/lo planes written TRANSPOSED [B, H, 64, ld] (column 1 + token): 32 consecutive tokens of a lane half are 64 contiguous bytes.
        const int HD = g.epi_heads * 64;
        const int ncol = n0 + wn * 64 - (qkv ? HD : 0);   // (q | k | v: the k / v columns start behind the H 64 query columns)
        const bool is_v = ncol >= HD;
        const int head = (is_v ? ncol - HD : ncol) >> 6;
        const _Float16* aux = reinterpret_cast<const _Float16*>(g.epi_aux);
        unsigned bad = 0;
        f32x4 ksc4[2][4];   // (k_scale in front of the first store, as above)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) ksc4[j][qq] = *reinterpret_cast<const f32x4*>(g.epi_scale + j * 32 + 8 * qq + 4 * h);
        if constexpr (STG) {
            // The TRANSPOSED value planes [B, H, 64, ld]: the direct form is 128 two-byte store instructions per wave (32 tokens x 2 columns each) - the longest epilogue
            // of the step (12 us + 4 us until the CU is free again, tools/gemm_trace.py).  Staged: the patch goes to LDS transposed, [column d][1 + token] as packed
            // (hi | lo << 16) words - one column of padding in front because key position = 1 + token (position 0 of a batch element is the learned null value), so that
            // the 16-byte groups of the PLANE rows are 16-byte groups of the LDS rows - and leaves as 8 positions per lane: one 16-byte store per plane for seven lanes of
            // a column, the first lane its 7 real positions in pieces (or all 8 when position 0 is the null value), one lane the patch's last token.  80 store
            // instructions per wave instead of 128, 1400 lane-stores instead of 8192.  Needs the patch inside one batch element and the problem (epi_rows % 64 == 0,
            // epi_ld % 8 == 0, no ragged last tile); otherwise the direct form below.
            const int prow0 = m0 + wm * WROWS;
            if (g.row_major_epi && is_v && (g.epi_rows % WROWS) == 0 && (g.epi_ld & 7) == 0 && prow0 + WROWS <= g.M) {
                unsigned* plu = reinterpret_cast<unsigned*>(pl);
#pragma unroll
                for (int i = 0; i < TI; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int q = 0; q < 16; ++q) {
                            const float val = (accM[i][j][q] + accC[i][j][q] * kGLoInv) * g.alpha;
                            const _Float16 hi = split_hi(val);
                            const _Float16 lo = split_lo(val, hi);
                            const int d = j * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
                            plu[d * VT_LD + 1 + i * 32 + r] = (unsigned)__builtin_bit_cast(unsigned short, hi) | ((unsigned)__builtin_bit_cast(unsigned short, lo) << 16);
                        }
                const int bb = prow0 / g.epi_rows, nk0 = prow0 - bb * g.epi_rows;   // (wave-uniform)
                _Float16* Vh = reinterpret_cast<_Float16*>(g.epi_hi2);
                _Float16* Vl = reinterpret_cast<_Float16*>(g.epi_lo2);
                constexpr int G8 = WROWS / 8;   // 16-byte position groups per column = lanes per column; 64 / G8 columns per pass
                const int gq = lane % G8;
#pragma unroll
                for (int ps = 0; ps < G8; ++ps) {
                    const int d = ps * (64 / G8) + lane / G8;
                    const u32x4 u0 = *reinterpret_cast<const u32x4*>(plu + d * VT_LD + 8 * gq);
                    const u32x4 u1 = *reinterpret_cast<const u32x4*>(plu + d * VT_LD + 8 * gq + 4);
                    u32x4 hw, lw;   // positions 8 gq .. 8 gq + 7 of column d: hi halves, lo halves
                    hw[0] = __builtin_amdgcn_perm(u0[1], u0[0], 0x05040100u); lw[0] = __builtin_amdgcn_perm(u0[1], u0[0], 0x07060302u);
                    hw[1] = __builtin_amdgcn_perm(u0[3], u0[2], 0x05040100u); lw[1] = __builtin_amdgcn_perm(u0[3], u0[2], 0x07060302u);
                    hw[2] = __builtin_amdgcn_perm(u1[1], u1[0], 0x05040100u); lw[2] = __builtin_amdgcn_perm(u1[1], u1[0], 0x07060302u);
                    hw[3] = __builtin_amdgcn_perm(u1[3], u1[2], 0x05040100u); lw[3] = __builtin_amdgcn_perm(u1[3], u1[2], 0x07060302u);
                    const long rowp = (((long)bb * g.epi_heads + head) * 64 + d) * g.epi_ld + nk0 + 8 * gq;   // position nk0 + 8 gq of this column's plane row
                    if (gq == 0) {
                        if (nk0 == 0) {   // position 0 = the learned null value of this (batch, head)
                            hw[0] = (hw[0] & 0xFFFF0000u) | (unsigned)__builtin_bit_cast(unsigned short, aux[2 * HD + head * 64 + d]);
                            lw[0] = (lw[0] & 0xFFFF0000u) | (unsigned)__builtin_bit_cast(unsigned short, aux[3 * HD + head * 64 + d]);
                            bad |= f16x2_nonfinite(hw[0] & 0xFFFF0000u) | f16x2_nonfinite(hw[1]) | f16x2_nonfinite(hw[2]) | f16x2_nonfinite(hw[3]);
                            *reinterpret_cast<u32x4*>(Vh + rowp) = hw;
                            *reinterpret_cast<u32x4*>(Vl + rowp) = lw;
                        } else {          // position nk0 belongs to the patch above: 7 positions in pieces of 2, 4 and 8 bytes
                            bad |= f16x2_nonfinite(hw[0] & 0xFFFF0000u) | f16x2_nonfinite(hw[1]) | f16x2_nonfinite(hw[2]) | f16x2_nonfinite(hw[3]);
                            *reinterpret_cast<unsigned short*>(Vh + rowp + 1) = (unsigned short)(hw[0] >> 16);
                            *reinterpret_cast<unsigned short*>(Vl + rowp + 1) = (unsigned short)(lw[0] >> 16);
                            *reinterpret_cast<unsigned*>(Vh + rowp + 2) = hw[1];
                            *reinterpret_cast<unsigned*>(Vl + rowp + 2) = lw[1];
                            *reinterpret_cast<uint2*>(Vh + rowp + 4) = make_uint2(hw[2], hw[3]);
                            *reinterpret_cast<uint2*>(Vl + rowp + 4) = make_uint2(lw[2], lw[3]);
                        }
                    } else {
                        bad |= f16x2_nonfinite(hw[0]) | f16x2_nonfinite(hw[1]) | f16x2_nonfinite(hw[2]) | f16x2_nonfinite(hw[3]);
                        *reinterpret_cast<u32x4*>(Vh + rowp) = hw;
                        *reinterpret_cast<u32x4*>(Vl + rowp) = lw;
                        if (gq == 1) {    // the patch's last token: position nk0 + WROWS
                            const unsigned ut = plu[d * VT_LD + WROWS];
                            bad |= f16x2_nonfinite(ut & 0xFFFFu);
                            const long tp = (((long)bb * g.epi_heads + head) * 64 + d) * g.epi_ld + nk0 + WROWS;
                            *reinterpret_cast<unsigned short*>(Vh + tp) = (unsigned short)(ut & 0xFFFFu);
                            *reinterpret_cast<unsigned short*>(Vl + tp) = (unsigned short)(ut >> 16);
                        }
                    }
                }
                if (bad) status_raise(g.status, BG_ST_F16_RANGE);
                return;
            }
            if (g.row_major_epi && !is_v) {   // the key planes
                _Float16* Kh = reinterpret_cast<_Float16*>(g.epi_hi);
                _Float16* Kl = reinterpret_cast<_Float16*>(g.epi_lo);
#pragma unroll
                for (int i = 0; i < TI; ++i) {
                    float v[2][16];
                    float ss = 0.f;
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int q = 0; q < 16; ++q) {
                            v[j][q] = (accM[i][j][q] + accC[i][j][q] * kGLoInv) * g.alpha;
                            ss = fmaf(v[j][q], v[j][q], ss);
                        }
                    ss += xor32(ss);
                    const float nrm = fmaxf(sqrtf(ss), 1e-12f);
                    const RowDiv rdiv(nrm);
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int qq = 0; qq < 4; ++qq) {
                            const f32x4 sc = ksc4[j][qq];
                            f32x4 o;
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[e] = rdiv(v[j][qq * 4 + e]) * sc[e];
                            *reinterpret_cast<f32x4*>(pl + (i * 32 + r) * RS + j * 32 + 8 * qq + 4 * h) = o;
                        }
                }
#pragma unroll
                for (int ps = 0; ps < WROWS / 8; ++ps) {
                    const int row = ps * 8 + rm_row, m = m0 + wm * WROWS + row;
                    f32x4 a0, a1;
                    stage_read8(row, a0, a1);
                    half8 hi8, lo8;
                    split8(a0, a1, hi8, lo8);
                    if (m < g.M) {
                        guard_half8(hi8, bad);
                        const int bb = m / g.epi_rows, nk = m - bb * g.epi_rows;
                        const long dst = (((long)bb * g.epi_heads + head) * g.epi_ld + 1 + nk) * 64 + 8 * rm_c8;
                        *reinterpret_cast<half8*>(Kh + dst) = hi8;
                        *reinterpret_cast<half8*>(Kl + dst) = lo8;
                        if (nk == 0) {   // the learned null key of this (batch, head): row 0
                            *reinterpret_cast<half8*>(Kh + dst - 64) = *reinterpret_cast<const half8*>(aux + head * 64 + 8 * rm_c8);
                            *reinterpret_cast<half8*>(Kl + dst - 64) = *reinterpret_cast<const half8*>(aux + HD + head * 64 + 8 * rm_c8);
                        }
                    }
                }
                if (bad) status_raise(g.status, BG_ST_F16_RANGE);
                return;
            }
        }
#pragma unroll
        for (int i = 0; i < TI; ++i) {
            const int m = m0 + wm * WROWS + i * 32 + r;
            const int mc = min(m, g.M - 1);
            const int bb = mc / g.epi_rows, nk = mc - bb * g.epi_rows;
            float v[2][16];
            float ss = 0.f;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    v[j][q] = (accM[i][j][q] + accC[i][j][q] * kGLoInv) * g.alpha;
                    ss = fmaf(v[j][q], v[j][q], ss);
                }
            ss += xor32(ss);   // (uniform control flow up to here: the exchange needs both lane halves)
            if (m >= g.M) continue;
            if (!is_v) {
                const float nrm = fmaxf(sqrtf(ss), 1e-12f);
                    const RowDiv rdiv(nrm);
                _Float16* Kh = reinterpret_cast<_Float16*>(g.epi_hi);
                _Float16* Kl = reinterpret_cast<_Float16*>(g.epi_lo);
                const long dst = (((long)bb * g.epi_heads + head) * g.epi_ld + 1 + nk) * 64;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) {
                        const int d = j * 32 + 8 * qq + 4 * h;
                        const f32x4 sc = ksc4[j][qq];
                        half4_t hi4, lo4;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float kv = rdiv(v[j][qq * 4 + e]) * sc[e];
                            hi4[e] = split_hi(kv);
                            lo4[e] = split_lo(kv, hi4[e]);
                        }
                        guard_half4(hi4, bad);
                        *reinterpret_cast<half4_t*>(Kh + dst + d) = hi4;
                        *reinterpret_cast<half4_t*>(Kl + dst + d) = lo4;
                        if (nk == 0) {   // the learned null key of this (batch, head): row 0
                            *reinterpret_cast<half4_t*>(Kh + dst - (long)64 + d) = *reinterpret_cast<const half4_t*>(aux + head * 64 + d);
                            *reinterpret_cast<half4_t*>(Kl + dst - (long)64 + d) = *reinterpret_cast<const half4_t*>(aux + HD + head * 64 + d);
                        }
                    }
            } else {
                _Float16* Vh = reinterpret_cast<_Float16*>(g.epi_hi2);
                _Float16* Vl = reinterpret_cast<_Float16*>(g.epi_lo2);
                const long base = ((long)bb * g.epi_heads + head) * 64 * g.epi_ld + 1 + nk;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const int d = j * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
                        const _Float16 hi = split_hi(v[j][q]);
                        guard_half(hi, bad);
                        Vh[base + (long)d * g.epi_ld] = hi;
                        Vl[base + (long)d * g.epi_ld] = split_lo(v[j][q], hi);
                        if (nk == 0) {
                            Vh[base - 1 + (long)d * g.epi_ld] = aux[2 * HD + head * 64 + d];
                            Vl[base - 1 + (long)d * g.epi_ld] = aux[3 * HD + head * 64 + d];
                        }
                    }
            }
        }
        if (bad) status_raise(g.status, BG_ST_F16_RANGE);
        return;
    }
    if (MODE == MODE_PLAIN && g.epi == EPI_GEGLU) {   // (compiled out of the convolution variant: its register budget has no room for a third epilogue)
        // GEGLU fused into the feed-forward up-projection (muse_net:71-76: x, gate = chunk(2); gate * gelu(x)): the weight rows were ordered so that a
        // wave's MFMA column tile j = 0 holds 32 `x` columns and j = 1 the 32 matching `gate` columns; the product leaves as ONE [M, N/2] matrix
        // (half the bytes of the raw projection, and the separate GEGLU pass over [M, N] disappears: its LayerNorm half runs on the result).
        float* C = g.C;
        _Float16* Pp = reinterpret_cast<_Float16*>(g.ln_out_planes);   // folded LayerNorm, producer side: raw planes + per-(row, 32 columns) statistics instead of fp32 C
        unsigned bad = 0;
        if constexpr (STG) {
            if (g.row_major_epi && Pp) {
                // the wave's 32 outputs of a row are ONE 128-byte line of the plane image ([hi 32 | lo 32] halves): staged [64 rows][32 outputs], then 8 lanes per row
                // (4 x 8 hi halves, 4 x 8 lo halves) write the whole line, 8 rows per instruction.  Statistics on the accumulator side, as in the direct form
#pragma unroll
                for (int i = 0; i < TI; ++i) {
                    const int m = m0 + wm * WROWS + i * 32 + r;
                    const int mc = min(m, g.M - 1);
                    float s1 = 0.f, s2 = 0.f;
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) {
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int q = qq * 4 + e;
                            const float xa = (accM[i][0][q] + accC[i][0][q] * kGLoInv) * g.alpha;
                            const float gt = (accM[i][1][q] + accC[i][1][q] * kGLoInv) * g.alpha;
                            v[e] = gt * gelu_erf(xa);
                        }
                        s1 += (v[0] + v[1]) + (v[2] + v[3]);
                        s2 += fmaf(v[0], v[0], v[1] * v[1]) + fmaf(v[2], v[2], v[3] * v[3]);
                        *reinterpret_cast<f32x4*>(pl + (i * 32 + r) * RS + 8 * qq + 4 * h) = v;
                    }
                    s1 += xor32(s1); s2 += xor32(s2);
                    if (h == 0 && m < g.M) reinterpret_cast<float2*>(g.ln_out_stats)[(long)((n0 >> 6) + wn) * g.ln_rows + mc] = make_float2(s1, s2);
                }
                const int c4 = rm_c8 & 3, lo_half = rm_c8 >> 2;   // lanes 0-3 of a row: the hi halves of outputs 8 c4 .. 8 c4 + 7, lanes 4-7 the lo halves
                const int c32 = (n0 >> 1) + wn * 32;              // first output column of the wave (a multiple of 32)
#pragma unroll
                for (int ps = 0; ps < WROWS / 8; ++ps) {
                    const int row = ps * 8 + rm_row, m = m0 + wm * WROWS + row;
                    const f32x4 a0 = *reinterpret_cast<const f32x4*>(pl + row * RS + 8 * c4);
                    const f32x4 a1 = *reinterpret_cast<const f32x4*>(pl + row * RS + 8 * c4 + 4);
                    half8 hi8, lo8;
                    split8(a0, a1, hi8, lo8);
                    if (m < g.M) {
                        if (!lo_half) guard_half8(hi8, bad);
                        *reinterpret_cast<half8*>(Pp + (long)m * 2 * g.ln_out_ld + (c32 >> 5) * 64 + 8 * c4 + (lo_half ? 32 : 0)) = lo_half ? lo8 : hi8;
                    }
                }
                if (bad) status_raise(g.status, BG_ST_F16_RANGE);
                return;
            }
        }
#pragma unroll
        for (int i = 0; i < TI; ++i) {
            const int m = m0 + wm * WROWS + i * 32 + r;
            const int mc = min(m, g.M - 1);
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const int o = (n0 >> 1) + wn * 32 + 8 * qq + 4 * h;   // output column: this tile's 64 outputs start at n0 / 2
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int q = qq * 4 + e;
                    const float xa = (accM[i][0][q] + accC[i][0][q] * kGLoInv) * g.alpha;
                    const float gt = (accM[i][1][q] + accC[i][1][q] * kGLoInv) * g.alpha;
                    v[e] = gt * gelu_erf(xa);
                }
                if (Pp) {
                    // 16-byte plane stores: the lane halves of a row hold neighbouring 4-column groups - the lower half stores the hi parts of the pair's 8 columns,
                    // the upper half the lo parts (one exchange of two registers; the layernorm kernel's store shape)
                    s1 += (v[0] + v[1]) + (v[2] + v[3]);
                    s2 += fmaf(v[0], v[0], v[1] * v[1]) + fmaf(v[2], v[2], v[3] * v[3]);
                    half4_t hi4, lo4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { hi4[e] = split_hi(v[e]); lo4[e] = split_lo(v[e], hi4[e]); }
                    guard_half4(hi4, bad);
                    const uint2 hw = __builtin_bit_cast(uint2, hi4), lw = __builtin_bit_cast(uint2, lo4);
                    const uint2 send = h ? hw : lw;
                    const uint2 recv = make_uint2(__float_as_uint(xor32(__uint_as_float(send.x))), __float_as_uint(xor32(__uint_as_float(send.y))));
                    const uint4 out = h ? make_uint4(recv.x, recv.y, lw.x, lw.y) : make_uint4(hw.x, hw.y, recv.x, recv.y);
                    const int c8 = (n0 >> 1) + wn * 32 + 8 * qq;   // first of the pair's 8 columns
                    if (m < g.M) *reinterpret_cast<uint4*>(Pp + (long)m * 2 * g.ln_out_ld + (c8 >> 5) * 64 + (c8 & 31) + (h ? 32 : 0)) = out;
                } else if (m < g.M && C) {
                    *reinterpret_cast<f32x4*>(C + (long)m * g.ldc + o) = v;
                }
            }
            if (Pp) {   // (uniform: every lane takes part in the exchange)
                s1 += xor32(s1); s2 += xor32(s2);
                if (h == 0 && m < g.M) reinterpret_cast<float2*>(g.ln_out_stats)[(long)((n0 >> 6) + wn) * g.ln_rows + mc] = make_float2(s1, s2);
            }
        }
        if (bad) status_raise(g.status, BG_ST_F16_RANGE);
        return;
    }
    if (KS && ksl > 1) {   // raw tile sums of this k slice; launch_splitk_reduce adds the slices in order and applies the epilogue
        float* P = g.kpart + (long)kz * g.M * g.N;
#pragma unroll
        for (int i = 0; i < TI; ++i) {
            const int m = m0 + wm * WROWS + i * 32 + r;
            if (m >= g.M) continue;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    const int n = n0 + wn * 64 + j * 32 + 8 * qq + 4 * h;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (n + e < g.N) P[(long)m * g.N + n + e] = accM[i][j][qq * 4 + e] + accC[i][j][qq * 4 + e] * kGLoInv;
                }
        }
        return;
    }
    }
    float* C = g.C;
    const float* Rp = g.R;
    unsigned ln_bad = 0;
    const bool vec_ok = ((g.ldc & 3) == 0) && (!Rp || (g.ldr & 3) == 0) && ((reinterpret_cast<uintptr_t>(C) & 15) == 0) &&
                        (!Rp || (reinterpret_cast<uintptr_t>(Rp) & 15) == 0);
    // Row-major store form (the throughput instantiation).  In the accumulator layout a store instruction covers 32 rows x 32 bytes: 32 PARTIAL lines, and a CU pushes
    // at most ~34 GB/s that way however idle the rest of the chip is (tools/storebw: 128 KB per tile = 3.8 us; sixteen lanes x 16 bytes = 256 contiguous bytes of one
    // row, four rows per instruction: 118 GB/s).  The phase stamps of the kernel (tools/gemm_trace.py) showed 6.5 us between the end of the k loop and the last store
    // issue of every round, synchronised or not.  So the wave's 64 x 64 patch goes through its own 17 KB of the (now dead) stage ring - written in the accumulator
    // layout, read back row-major; row stride 68 floats: both directions conflict-free for 16-lane groups of 16-byte accesses - and bias / activation / residual follow on
    // the row-major side, element by element in the same order as below (bit-identical).  No barrier: after the last k-tile's barrier nobody reads operand data from the
    // ring any more (the stale look-ahead fetch of the last iteration is never used), and a wave only touches its own slice.
    if constexpr (STG) {
        if (g.row_major_epi && vec_ok && (g.N & 3) == 0 && !g.ln_out_planes) {
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) {
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = accM[i][j][qq * 4 + e] + accC[i][j][qq * 4 + e] * kGLoInv;
                        *reinterpret_cast<f32x4*>(pl + (i * 32 + r) * RS + j * 32 + 8 * qq + 4 * h) = v;
                    }
            constexpr int LPR = WCOLS / 4, RPP = 64 / LPR, NP = WROWS / RPP;   // lanes per row (16 bytes each), rows per pass, passes
            const int c = lane % LPR, rr = lane / LPR;
            const int n = n0 + wn * WCOLS + 4 * c;
            const int nc = min(n, g.N - 4);
            f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
            if (g.bias_n) b4 = *reinterpret_cast<const f32x4*>(g.bias_n + nc);
            f32x4 rres[NP];
            float bmr[NP];
#pragma unroll
            for (int s2 = 0; s2 < NP; ++s2) {
                const int m = min(m0 + wm * WROWS + s2 * RPP + rr, g.M - 1);
                bmr[s2] = g.bias_m ? g.bias_m[m] : 0.f;
                rres[s2] = Rp ? *reinterpret_cast<const f32x4*>(Rp + (long)m * g.ldr + nc) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int s2 = 0; s2 < NP; ++s2) {
                const int row = s2 * RPP + rr, m = m0 + wm * WROWS + row;
                const f32x4 a = *reinterpret_cast<const f32x4*>(pl + row * RS + 4 * c);
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float t = a[e] * g.alpha + bmr[s2];
                    t += b4[e];
                    if (g.act == ACT_GELU) t = gelu_erf(t);
                    o[e] = t + rres[s2][e];
                }
                if (m < g.M && n < g.N) *reinterpret_cast<f32x4*>(C + (long)m * g.ldc + n) = o;
            }
            return;
        }
    }
    // Every load of the epilogue is issued BEFORE its first store.  Loads and stores share one counter on this part (vmcnt) and the compiler cannot tell that the
    // residual does not alias C (it usually IS C: x = x + proj), so with the loads inside the store loop every one of the sixteen (row tile, column quad) steps was
    // "load R; s_waitcnt vmcnt(0); store C" - a wait for the PREVIOUS step's stores to complete each time, ~0.7 us of L2 round trip sixteen times per tile with every
    // matrix pipe idle (the bias: four dword loads, each with its own wait).  Hoisted: one round trip for the tile, then the stores back to back.
    f32x4 bias4[TJ][4];
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            const int n = n0 + wn * WCOLS + j * 32 + 8 * qq + 4 * h;
#pragma unroll
            for (int e = 0; e < 4; ++e) bias4[j][qq][e] = (g.bias_n && n + e < g.N) ? g.bias_n[n + e] : 0.f;
        }
    f32x4 rv[TI][TJ][4];
    float bm[TI];
#pragma unroll
    for (int i = 0; i < TI; ++i) {
        const int m = min(m0 + wm * WROWS + i * 32 + r, g.M - 1);
        bm[i] = g.bias_m ? g.bias_m[m] : 0.f;
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const int n = n0 + wn * WCOLS + j * 32 + 8 * qq + 4 * h;
                if (Rp && vec_ok && n + 3 < g.N) rv[i][j][qq] = *reinterpret_cast<const f32x4*>(Rp + (long)m * g.ldr + n);
                else if (Rp) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) rv[i][j][qq][e] = Rp[(long)m * g.ldr + min(n + e, g.N - 1)];
                } else rv[i][j][qq] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
    }
#pragma unroll
    for (int i = 0; i < TI; ++i) {
        const int m = m0 + wm * WROWS + i * 32 + r;
        if (m >= g.M) continue;
        float ln_s1[TJ], ln_s2[TJ];
#pragma unroll
        for (int j = 0; j < TJ; ++j) { ln_s1[j] = 0.f; ln_s2[j] = 0.f; }
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const int n = n0 + wn * WCOLS + j * 32 + 8 * qq + 4 * h;
                if (n >= g.N) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int q = qq * 4 + e;
                    float t = (accM[i][j][q] + accC[i][j][q] * kGLoInv) * g.alpha + bm[i];
                    t += bias4[j][qq][e];
                    if (g.act == ACT_GELU) t = gelu_erf(t);
                    v[e] = t + rv[i][j][qq][e];
                }
                float* cp = C + (long)m * g.ldc + n;
                if (vec_ok && n + 3 < g.N) {
                    f32x4 o; o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3];
                    *reinterpret_cast<f32x4*>(cp) = o;
                    if (MODE == MODE_PLAIN && g.ln_out_planes) {
                        // folded LayerNorm, producer side (the residual-stream projections): besides the fp32 row the residual needs, the raw (hi, lo) planes the next
                        // projection reads and this lane's share of the row's (sum, sum of squares) over the 32 columns of MFMA tile j (completed below)
                        ln_s1[j] += (v[0] + v[1]) + (v[2] + v[3]);
                        ln_s2[j] += fmaf(v[0], v[0], v[1] * v[1]) + fmaf(v[2], v[2], v[3] * v[3]);
                        half4_t hi4, lo4;
#pragma unroll
                        for (int e = 0; e < 4; ++e) { hi4[e] = split_hi(v[e]); lo4[e] = split_lo(v[e], hi4[e]); }
                        guard_half4(hi4, ln_bad);
                        // 16-byte stores: the lane halves of a row hold neighbouring 4-column groups; the lower half stores the hi parts of the pair's 8 columns, the upper
                        // half the lo parts (both halves of a row are active together: m depends on lane & 31 only; the launcher guarantees N % 8 == 0)
                        const uint2 hw = __builtin_bit_cast(uint2, hi4), lw = __builtin_bit_cast(uint2, lo4);
                        const uint2 send = h ? hw : lw;
                        const uint2 recv = make_uint2(__float_as_uint(xor32(__uint_as_float(send.x))), __float_as_uint(xor32(__uint_as_float(send.y))));
                        const uint4 pk = h ? make_uint4(recv.x, recv.y, lw.x, lw.y) : make_uint4(hw.x, hw.y, recv.x, recv.y);
                        const int c8 = n - 4 * h;
                        *reinterpret_cast<uint4*>(reinterpret_cast<_Float16*>(g.ln_out_planes) + (long)m * 2 * g.ln_out_ld + (c8 >> 5) * 64 + (c8 & 31) + (h ? 32 : 0)) = pk;
                    }
                    if (CONV && g.gn_part) {
                        // GroupNorm statistics of the output tensor: this lane's 4 consecutive channels of its row, summed over the 32 rows (lanes) of the half -
                        // one (sum, sum of squares) pair per (32 rows, 4 channels).  The launcher guarantees whole tiles (every lane here, no tail), so the DPP row sums
                        // and the cross-row exchange run with all lanes active
                        float s4 = (v[0] + v[1]) + (v[2] + v[3]);
                        float q4 = fmaf(v[0], v[0], v[1] * v[1]) + fmaf(v[2], v[2], v[3] * v[3]);
                        s4 = row16_sum(s4); q4 = row16_sum(q4);
                        s4 += xor16(s4); q4 += xor16(q4);
                        if (r == 0) *reinterpret_cast<float2*>(g.gn_part + ((long)(m >> 5) * (g.N >> 2) + (n >> 2)) * 2) = make_float2(s4, q4);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (n + e < g.N) cp[e] = v[e];
                }
            }
            if (MODE == MODE_PLAIN && g.ln_out_planes) {   // the row's pair for the 32 columns of MFMA tile j: the two lane halves hold 16 columns each
                const float t1 = ln_s1[j] + xor32(ln_s1[j]), t2 = ln_s2[j] + xor32(ln_s2[j]);
                const int grp = (n0 + wn * WCOLS + j * 32) >> 5;
                if (h == 0 && grp * 32 < g.N) reinterpret_cast<float2*>(g.ln_out_stats)[(long)grp * g.ln_rows + m] = make_float2(t1, t2);
            }
        }
    }
    if (MODE == MODE_PLAIN && ln_bad) status_raise(g.status, BG_ST_F16_RANGE);
}


template <int MODE, int WM, int S, bool W16 = false, bool KS = false, int TI = 2, int TJ = 2>
__global__ __launch_bounds__(WM * 512 / (TI * TJ), (WM == 2 && S == 2) ? 2 : 1) void gemm_split_glds_kernel(GemmArgs g) {
    int tx, ty;
    if (g.tile_band > 0) xcd_tile_banded(gridDim.x, gridDim.y, g.tile_band, tx, ty);
    else xcd_tile(gridDim.x, gridDim.y, tx, ty);
    // split-K (gridDim.z slices, small-M problems): this block owns k-tiles [kt_first, kt_first + nk)
    const int nk_all = g.K / GBK, ksl = KS ? (int)gridDim.z : 1, kz = KS ? (int)blockIdx.z : 0;
    const int kt_first = kz * (nk_all / ksl) + min(kz, nk_all % ksl);
    const int nk = nk_all / ksl + (kz < nk_all % ksl ? 1 : 0);
#ifdef BEVGEN_GEMM_TRACE
    unsigned long long gt[5] = {};
    gt[0] = __builtin_amdgcn_s_memrealtime();
    gemm_tile<MODE, WM, S, W16, KS, TI, TJ>(g, tx, ty, kt_first, nk, ksl, kz, 0, 0, (int)threadIdx.x, gt);
    if (MODE == MODE_PLAIN && !KS) {
        gt[3] = __builtin_amdgcn_s_memrealtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        gt[4] = __builtin_amdgcn_s_memrealtime();
        const int lin = blockIdx.y * gridDim.x + blockIdx.x;
        if (threadIdx.x == 0 && lin < 2048) {
            unsigned xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            unsigned hwid;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            unsigned long long* rec = g_gemm_trace + ((long)g.epi * 2048 + lin) * 8;
            for (int i = 0; i < 5; ++i) rec[i] = gt[i];
            rec[5] = ((unsigned long long)xcc << 32) | hwid;
            rec[6] = ((unsigned long long)gridDim.x << 32) | gridDim.y;
            rec[7] = ((unsigned long long)(unsigned)g.K << 32) | (g.R ? 1u : 0u) | (g.ln_in_stats || g.ln_in_gsums ? 2u : 0u) | (g.ln_out_planes ? 4u : 0u);
        }
    }
#else
    gemm_tile<MODE, WM, S, W16, KS, TI, TJ>(g, tx, ty, kt_first, nk, ksl, kz, 0, 0, (int)threadIdx.x);
#endif
}

// Stream-K form for problems whose tile count does not fill whole rounds of the chip (one or two scenes: 144 / 48 / 258 tiles of 256 x 128 on 256 CUs).  The work is cut
// into UNITS of one k-tile of one block tile, numbered tile-major / k-minor; gridDim.x persistent workgroups each take a contiguous, equally long range of units.  A range
// covers at most: the HIGH-k end of one tile (whose low-k part lies with lower-indexed workgroups), some whole tiles, and the LOW-k start of one tile.  The range is walked
// from its top down: the low-k start first - its raw sums go to the workgroup's slot and its flag is raised EARLY - and the high-k end last: by then the lower-indexed
// contributors of that tile have long published theirs, and this workgroup (the "owner": it holds the last k part) adds them in ascending k order and runs the fused
// epilogue.  Sums are added in a fixed order: results are run-to-run identical; they differ from the one-workgroup-per-tile launch only in fp32 association.
template <int WM, int S, bool W16>
__global__ __launch_bounds__(WM * 128, 1) void gemm_split_glds_sk_kernel(GemmArgs g) {
    // workgroup b runs on XCD b % 8 (checked per device: xcd_placement_verified): XCD x takes the whole tiles [x T / 8, (x + 1) T / 8) and its gridDim.x / 8 workgroups
    // (local index b / 8) cut those tiles' units evenly - every exchange of partial sums stays inside one L2
    const int nk = g.K / GBK, gx = (g.N + GBN - 1) / GBN;
    const int x = blockIdx.x & 7, jl = blockIdx.x >> 3, Gx = gridDim.x >> 3;
    const long T = g.sk_tiles, tb = x * T / 8, te = (x + 1) * T / 8;
    const long Ux = (te - tb) * nk, u0 = tb * nk;
    const long ub = u0 + (long)jl * Ux / Gx, ue = u0 + ((long)jl + 1) * Ux / Gx;
    if (ue <= ub) return;
    for (long t = (ue - 1) / nk; t >= ub / nk; --t) {
        const int k0 = (int)(max(ub, t * nk) - t * nk), k1 = (int)(min(ue, (t + 1) * nk) - t * nk);
        int role = 0, first = 0;
        if (k0 != 0 || k1 != nk) {
            role = k1 == nk ? 2 : 1;
            if (role == 2) {   // the contributors: every lower workgroup of this XCD whose range reaches into this tile
                int fj = jl;
                while (fj > 0 && u0 + (long)fj * Ux / Gx > t * nk) --fj;
                first = 8 * fj + x;
            }
        }
        // (the thread index is made opaque per segment: otherwise every per-lane address of the tile body - loop-invariant in this loop - is hoisted out of it and
        // stays live across the whole body: 528 spilled VGPRs instead of none)
        int tid = (int)threadIdx.x;
        asm volatile("" : "+v"(tid));
        gemm_tile<MODE_PLAIN, WM, S, W16, false, 2, 2, true>(g, (int)(t % gx), (int)(t / gx), k0, k1 - k0, 1, 0, role, first, tid
#ifdef BEVGEN_GEMM_TRACE
                                                             , g_gemm_tr_tmp
#endif
                                                             );
        __syncthreads();   // the next segment's first DMA overwrites stages the slowest wave may still be reading
    }
}

// (Round 6 also measured a PERSISTENT whole-tile form - one workgroup per CU walking the tiles of its dispatch slot, optionally with a start offset per workgroup so that
// the 256 epilogue store bursts do not coincide: slower than letting the hardware dispatch a workgroup per tile at every shape (sixteen scenes, [24576, 1024] x 1024: 153 ->
// 169 us; x 32: 39 -> 57 us, i.e. +6 us per round of tiles; offsets of 1.5 / 3 us changed nothing) - removed; profiles/r06_ab_gemm_persistent.txt.  What the same probe shows
// about the throughput kernel: 35 us of fixed time per three-round launch (ring fill + epilogue drain, ~12 us per round) + 3.92 us per k-tile = 411 TF-equiv asymptotically,
// 0.39 of the ceiling at K = 1024.)

#ifdef BEVGEN_GEMM_TRACE
}  // namespace bevgen
extern "C" __attribute__((visibility("default"))) int bevgen_debug_gemm_trace(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(bevgen::g_gemm_trace), sizeof(bevgen::g_gemm_trace));
}
namespace bevgen {
#endif
size_t gemm_sk_ws_bytes() { return 1024 * sizeof(float) + (size_t)256 * (8 * 4 * 16 * 64) * sizeof(float); }   // 1024 flag words + 256 workgroup slots of 128 KiB

// Does the stream-K form pay?  T tiles of 256 x 128 cost ceil(T / 256) rounds of the chip; the form removes the empty part of the last round (and the second launch of a
// row-split problem) at the price of one partial-tile exchange per workgroup (~3 us) - worth it when at least 15 % of the rounds would be empty and every workgroup still
// gets a few k-tiles.  One scene (rows 1536): q|k|v 144 tiles (44 % empty), the 1024-wide projections 48 (81 %), the up-projection 258 (50 % of two rounds); two scenes:
// 288 / 96 / 516; sixteen scenes: 2304 = 9 rounds exactly, 768 = 3, 4128 = 16.1 (its row-split form stays).
bool gemm_sk_pays(long rows, int N, int K) {
    static const int sk_env = getenv("BEVGEN_GEMM_SK") ? atoi(getenv("BEVGEN_GEMM_SK")) : 0;   // (default off: see launch_gemm_split_glds)
    if (!sk_env) return false;
    const long T = (long)cdiv(rows, 256) * cdiv(N, GBN), rounds = (T + 255) / 256;
    const double empty = 1.0 - (double)T / (double)(rounds * 256);
    // ... and only while a tile is shared by two or three workgroups (T >= 128): with fewer tiles every tile's last workgroup merges five or more 128 KiB partials while
    // the others idle - measured slower than the 64-row blocks of the data-parallel launch at every such shape (profiles/r06_ab_gemm_sk_ops.txt)
    (void)K;
    return T >= 128 && empty >= 0.15;
}

void launch_gemm_split_glds(const GemmArgs& g_in, hipStream_t stream) {
    GemmArgs g = g_in;
    if (g.mode == MODE_CONV3) {
        if (g.conv_stride == 0) g.conv_stride = 1;
        if (g.conv_pad < 0) g.conv_pad = 1;
        if (g.conv_hin == 0) g.conv_hin = g.conv_up ? g.conv_h / 2 : g.conv_h;
        if (g.conv_win == 0) g.conv_win = g.conv_up ? g.conv_w / 2 : g.conv_w;
        BG_REQUIRE(g.conv_cin % GBK == 0 && g.K == 9 * g.conv_cin, "conv3x3: Cin=%d must be a multiple of 32", g.conv_cin);
        const long a_bytes = (long)(g.M / (g.conv_h * g.conv_w)) * g.conv_hin * g.conv_win * g.conv_cin * 4;
        BG_REQUIRE(a_bytes < 0xFFFFFF00L, "conv3x3 (LDS-DMA): the activation planes (%ld bytes) must stay below 4 GiB per launch", a_bytes);
        g.a_bytes = (int)(unsigned)a_bytes;
    }
    BG_REQUIRE(g.A_hi && g.A_lo && g.B_hi && g.B_lo, "gemm_split_glds: both operands must be pre-split");
    g.status = status_current();
    static const int rpf_env = getenv("BEVGEN_GEMM_RPF") ? atoi(getenv("BEVGEN_GEMM_RPF")) : 1;   // residual prefetch in the k loop's tail (0: off, for A/B runs; profiles/r06_ab_gemm_rpf.txt)
    static const int rme_env = getenv("BEVGEN_GEMM_RME") ? atoi(getenv("BEVGEN_GEMM_RME")) : 1;   // row-major store form of the plain epilogue (0: accumulator-layout stores, A/B runs)
    g.row_major_epi = rme_env != 0;
    g.r_prefetch = rpf_env && g.mode == MODE_PLAIN && g.R && (g.ldr & 3) == 0 && (reinterpret_cast<uintptr_t>(g.R) & 15) == 0 && (long)g.M * g.ldr * 4 < 0x7FFFFFFFL && g.N >= 64;
    if (g.gn_part)
        BG_REQUIRE(g.mode == MODE_CONV3 && g.epi == 0 && g.ksplit <= 1 && g.M % 256 == 0 && g.m_base == 0 && g.N % GBN == 0 && g.ldc == g.N && (g.ldc & 3) == 0 &&
                       (!g.R || (g.ldr & 3) == 0) && (reinterpret_cast<uintptr_t>(g.C) & 15) == 0 && (!g.R || (reinterpret_cast<uintptr_t>(g.R) & 15) == 0) && !g.bias_m,
                   "gemm_split_glds: GroupNorm partials need whole 256 x 128 tiles of a convolution with the plain epilogue (M=%d N=%d)", g.M, g.N);
    BG_REQUIRE(g.K % GBK == 0 && g.lda % GBK == 0 && g.ldb % GBK == 0, "gemm_split_glds: K, lda, ldb must be multiples of 32 (K=%d lda=%d ldb=%d)", g.K, g.lda, g.ldb);
    BG_REQUIRE(g.batch == 1, "gemm_split_glds: batched form not provided");
    if (g.epi == EPI_MUSE_KV)
        BG_REQUIRE(g.mode == MODE_PLAIN && g.N == 2 * g.epi_heads * 64 && g.epi_hi && g.epi_lo && g.epi_hi2 && g.epi_lo2 && g.epi_aux && g.epi_scale && g.epi_rows > 0 &&
                       g.epi_ld >= g.epi_rows + 1 && !g.R && !g.bias_n && !g.bias_m && g.act == ACT_NONE && (g.no_row_split || g.M % g.epi_rows == 0),
                   "gemm_split_glds: bad fused k/v-preparation arguments");
    if (g.epi == EPI_GEGLU)
        BG_REQUIRE(g.mode == MODE_PLAIN && g.N % GBN == 0 && (g.ldc % 4 == 0 && g.ldc >= g.N / 2) && !g.R && !g.bias_n && !g.bias_m && g.act == ACT_NONE,
                   "gemm_split_glds: bad fused GEGLU arguments (N=%d ldc=%d)", g.N, g.ldc);
    if (g.epi == EPI_MUSE_QKV)
        BG_REQUIRE(g.mode == MODE_PLAIN && g.N == 3 * g.epi_heads * 64 && g.epi_hi && g.epi_lo && g.epi_hi2 && g.epi_lo2 && g.epi_aux && g.epi_scale && g.epi_qh && g.epi_ql &&
                       g.epi_qscale && g.epi_rows > 0 && g.epi_ld >= g.epi_rows + 1 && !g.R && !g.bias_n && !g.bias_m && g.act == ACT_NONE &&
                       (g.no_row_split || g.M % g.epi_rows == 0) && g.ksplit <= 1,
                   "gemm_split_glds: bad fused q/k/v-preparation arguments");
    if (g.epi == EPI_MUSE_Q)
        BG_REQUIRE(g.mode == MODE_PLAIN && g.N % 64 == 0 && g.epi_hi && g.epi_lo && g.epi_scale && g.epi_rows > 0 && g.epi_heads * 64 == g.N && !g.R && !g.bias_n && !g.bias_m,
                   "gemm_split_glds: bad fused q-preparation arguments");
    if (g.ln_in_stats || g.ln_in_gsums)
        BG_REQUIRE(g.mode == MODE_PLAIN && g.ln_in_cs && g.N % 4 == 0 && g.ksplit <= 1 && !(g.ln_in_stats && g.ln_in_gsums) &&
                       (!g.ln_in_gsums || (g.ln_in_groups > 0 && g.ln_in_count > 0 && g.ln_rows >= g.M)),
                   "gemm_split_glds: bad folded-LayerNorm consumer arguments (N=%d ksplit=%d groups=%d)", g.N, g.ksplit, g.ln_in_groups);
    if (g.ln_out_planes)
        BG_REQUIRE(g.mode == MODE_PLAIN && g.ln_out_stats && g.ln_rows >= g.M && g.ln_out_ld % 32 == 0 && g.ksplit <= 1 &&
                       (g.epi == EPI_GEGLU ? g.ln_out_ld * 2 >= g.N : (g.epi == 0 && g.ln_out_ld >= g.N && g.N % 32 == 0 && (g.ldc & 3) == 0 && (!g.R || (g.ldr & 3) == 0) &&
                                                                        (reinterpret_cast<uintptr_t>(g.C) & 15) == 0 && (!g.R || (reinterpret_cast<uintptr_t>(g.R) & 15) == 0))),
                   "gemm_split_glds: bad folded-LayerNorm producer arguments (ld=%d N=%d epi=%d)", g.ln_out_ld, g.N, g.epi);
    // ---- stream-K route (the caller provided a workspace: Route M's projections): problems whose 256 x 128 tiles would leave much of their last round of the chip empty
    // DEFAULT OFF: measured slower than the launcher's other choices at every Route-M shape of one, two and four scenes (profiles/r06_ab_gemm_sk_ops.txt: +4 .. +27 us per
    // projection; one scene 163.8 -> 168.5 ms with the routing rule below, 214 ms with every projection): a partial 256 x 128 tile is 128 KiB to publish and to read back,
    // every segment refills the three-stage ring, and the workgroup that holds a tile's last k range merges while its peers idle - together more than the empty part of
    // the last round they remove.  $BEVGEN_GEMM_SK=1 routes by gemm_sk_pays, 2 takes the form whenever a workspace is given (the operator tests force it per call)
    static const int sk_env = getenv("BEVGEN_GEMM_SK") ? atoi(getenv("BEVGEN_GEMM_SK")) : 0;
    if (g.sk_ws && (sk_env || g.sk_force) && g.mode == MODE_PLAIN && g.ksplit <= 1 && g.m_base == 0 && !g.no_row_split && !g.bias_m && !g.ln_in_gsums && xcd_placement_verified() &&
        (sk_env == 2 || g.sk_force || gemm_sk_pays(g.M, g.N, g.K))) {
        static std::atomic<int> cu_count[kMaxDevices];
        static std::atomic<bool> sk_attr[kMaxDevices];
        const int ds = device_slot();
        int cus = cu_count[ds].load(std::memory_order_acquire);
        if (cus == 0) {
            int dev = 0;
            HIP_CHECK(hipGetDevice(&dev));
            HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
            cu_count[ds].store(cus, std::memory_order_release);
        }
        const size_t lds_sk = (size_t)3 * 384 * 2 * GBK * 2 + 4096;
        if (!sk_attr[ds].load(std::memory_order_acquire)) {
            HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_split_glds_sk_kernel<4, 3, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_sk));
            HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_split_glds_sk_kernel<4, 3, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_sk));
            sk_attr[ds].store(true, std::memory_order_release);
        }
        g.sk_tiles = cdiv(g.M, 256) * cdiv(g.N, GBN);
        const long units = (long)g.sk_tiles * (g.K / GBK);
        const int G = (int)std::max<long>(8, std::min<long>(std::min(cus, 256), units / 2)) & ~7;   // (a multiple of 8: whole tiles per XCD; >= two k-tiles per workgroup; 256 slots)
        g.tile_band = 0;
        ProfScope prof(PROF_GEMM_SMALL, 2.0 * g.M * (double)g.N * g.K, stream);
        if (g.b_lo_zero) hipLaunchKernelGGL((gemm_split_glds_sk_kernel<4, 3, true>), dim3(G), dim3(512), lds_sk, stream, g);
        else hipLaunchKernelGGL((gemm_split_glds_sk_kernel<4, 3, false>), dim3(G), dim3(512), lds_sk, stream, g);
        LAUNCH_CHECK();
        return;
    }
    g.tile_band = 4;   // band height of the XCD-aware tile order (measured optimum for 256 x 128 tiles)
    static const int band_env = getenv("BEVGEN_GEMM_BAND") ? atoi(getenv("BEVGEN_GEMM_BAND")) : 0;   // A/B switch (tools/ab.sh m env BEVGEN_GEMM_BAND=2,4,8)
    if (band_env > 0) g.tile_band = band_env;
    static const int force_wm = getenv("BEVGEN_GEMM_WM") ? atoi(getenv("BEVGEN_GEMM_WM")) : 0;   // 2 | 4: pins the block rows (128 | 256) for A/B runs
    // 256-row tiles (8 waves, 3 stages, one block per CU) unless the problem is too small to give every CU one of them; then 128-row tiles
    // with 2 stages (64 KiB) so that two independent 4-wave blocks share a CU
    const int rows = g.M - g.m_base;   // (g.M is the END row of this launch, g.m_base its first)
    static const int top_wm_env = getenv("BEVGEN_GEMM_TOPWM") ? atoi(getenv("BEVGEN_GEMM_TOPWM")) : 4;
    const int wm = (force_wm == 2 || force_wm == 4) ? force_wm : (g.force_wm == 4 && top_wm_env == 4) ? 4 : ((long)cdiv(rows, 256) * cdiv(g.N, GBN) >= 256 ? 4 : 2);
    // Tile quantisation: T tiles of 256 x 128 on 256 CUs cost ceil(T / 256) rounds - the up-projection of sixteen scenes is 4128 tiles = 16.1 rounds and pays 17, of one
    // scene 258 tiles and pays 2, a [12288, 1024] projection of the three-camera shape 384 tiles and pays 2.  When the last round would hold at most 128 tiles, the launch
    // is cut at a row-tile boundary: the first part fills whole rounds, the rest (<= 128 tiles' worth of rows) runs as 128-row blocks, one short round of its own
    // (about 0.45 of a full one).  Same kernels, same per-row arithmetic: results are bit-identical to the single launch.  $BEVGEN_GEMM_ROWSPLIT=0 turns it off (A/B runs)
    static const int rowsplit_env = getenv("BEVGEN_GEMM_ROWSPLIT") ? atoi(getenv("BEVGEN_GEMM_ROWSPLIT")) : 1;
    if (rowsplit_env && !g.no_row_split && wm == 4 && g.mode == MODE_PLAIN && g.ksplit <= 1) {
        const long gx = cdiv(g.N, GBN), gy = cdiv(rows, 256), T = gx * gy;
        const long full = T / 256;                       // whole rounds
        const long gy_top = full * 256 / gx;             // row tiles that fit them
        const long rest = (gy - gy_top) * gx;            // tiles left for the last round
        if (T % 256 != 0 && full >= 1 && gy_top >= 1 && gy_top < gy && rest <= 128) {
            GemmArgs top = g, bot = g;
            top.no_row_split = bot.no_row_split = true;
            top.force_wm = 4;                            // the part that was sized to fill whole rounds of 256-row blocks keeps them
            top.M = g.m_base + (int)gy_top * 256;        // end row of the first part
            bot.m_base = top.M;
            // ... and the rest picks its block by its OWN size even when the caller pinned 256-row blocks for the whole problem (q | k | v of two scenes: 288 tiles = 240 +
            // 48; the 48 as 256-row blocks kept 48 CUs busy for 33.5 us, as 192 blocks of 64 rows ~18 us; profiles/r06_ab_rowsplit_bot.txt).  $BEVGEN_ROWSPLIT_BOT_WM=4: as before
            static const int bot_wm_env = getenv("BEVGEN_ROWSPLIT_BOT_WM") ? atoi(getenv("BEVGEN_ROWSPLIT_BOT_WM")) : 0;
            if (bot_wm_env != 4) bot.force_wm = 0;
            // (the rest on a side stream BESIDE a first part that leaves CUs idle - one scene: 215 blocks on 256 CUs - was measured and rejected: the fork / join events cost
            // more than the overlap buys, one scene 162.9 -> 173.4 ms, sixteen scenes 10.34 -> 10.30 scenes/s; profiles/r05_ab_rowsplit_side_*.txt)
            launch_gemm_split_glds(top, stream);
            launch_gemm_split_glds(bot, stream);
            return;
        }
    }
    // ... unless even those leave CUs without a second block (a batch of one or two scenes): then nothing shares the CU, and the block becomes eight waves with
    // 32x64 patches on a four-stage ring (three k-tiles in flight instead of one; the kernel's TI note).  Measured on the Route M step (tools/ab_env_m.sh): one scene
    // 236 -> 199 ms, two scenes 303 -> 270 ms; at three and four scenes (288 / 384 blocks: two four-wave blocks per CU) the eight-wave block is 2-3 % slower.
    // $BEVGEN_GEMM_STAGES = 2 | 8 | 16 pins the small-problem shape for A/B runs and tests (2: four waves, two stages; 8: eight waves, four stages; 16: 64-row blocks)
    static const int stages_env = getenv("BEVGEN_GEMM_STAGES") ? atoi(getenv("BEVGEN_GEMM_STAGES")) : 0;
    static const int conv_thin_env = getenv("BEVGEN_CONV_THIN") ? atoi(getenv("BEVGEN_CONV_THIN")) : 1;
    const bool lone = (g.mode == MODE_PLAIN || conv_thin_env) && (long)cdiv(g.N, GBN) * cdiv(rows, 128) * g.ksplit <= 256;
    // ... and when even the 128-row blocks cover at most half of the CUs (a [1536, 1024] projection: 96), 64-row blocks of four waves (32x64 patches, four stages): twice
    // the blocks, a shorter k-tile each (16 = that shape): 21.4 -> 18.4 us at K = 1024, one-scene step 195.7 -> 187.4 ms on the same box (profiles/r03_ab_b1_half_rows.txt)
    const bool half_rows = lone && g.mode == MODE_PLAIN && g.ksplit == 1 && (long)cdiv(g.N, GBN) * cdiv(rows, 128) <= 128;
    int shape = lone ? (half_rows ? 16 : 8) : 2;
    if (g.mode == MODE_PLAIN && (stages_env == 2 || stages_env == 8 || (stages_env == 16 && g.ksplit == 1))) shape = stages_env;
    const int stages = wm == 4 ? 3 : (shape == 2 ? 2 : 4);
    const bool thin = wm == 2 && shape == 8, half = wm == 2 && shape == 16;
    // ... and with the plain epilogue (bias / activation / residual: the fused ones need 64-column wave patches) the 64-row block runs on eight waves of 32x32 patches
    static const int half8_env = getenv("BEVGEN_GEMM_HALF8") ? atoi(getenv("BEVGEN_GEMM_HALF8")) : 1;
    const bool half8 = half && half8_env && g.epi == 0;
    const int tbm = half ? 64 : wm * 64;
    BG_REQUIRE(g.ksplit >= 1 && (g.ksplit == 1 || (g.kpart && g.epi == 0 && g.mode == MODE_PLAIN && wm == 2 && !half && g.K / GBK >= 2 * g.ksplit && !g.bias_m)),
               "gemm_split_glds: split-K needs a workspace, the plain epilogue, the 128-row tile and >= 2 k-tiles per slice (ksplit=%d K=%d)", g.ksplit, g.K);
    dim3 grid(cdiv(g.N, GBN), cdiv(rows, tbm), g.ksplit);
    const size_t lds = (size_t)stages * (tbm + GBN) * 2 * GBK * sizeof(_Float16) + ((g.ln_in_stats || g.ln_in_gsums || g.r_prefetch) ? 4096 : 0) + ((g.r_prefetch || g.ln_in_gsums) ? 2048 : 0) +
                       (g.ln_in_gsums ? 8192 : 0);   // LayerNorm (mean, rstd) slots | prefetch sink | the block merge's fp64 partial sums   // (+ the folded LayerNorm's per-row (mean, rstd) slots)
    static std::atomic<bool> attr_set[kMaxDevices];
    const int dslot = device_slot();
    if (!attr_set[dslot].load(std::memory_order_acquire)) {
#define BG_SET(K, BYTES) HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&K), hipFuncAttributeMaxDynamicSharedMemorySize, BYTES))
        BG_SET((gemm_split_glds_kernel<MODE_PLAIN, 2, 2>), 2 * 256 * 2 * GBK * 2 + 14336);
        BG_SET((gemm_split_glds_kernel<MODE_CONV3, 2, 2>), 2 * 256 * 2 * GBK * 2);
        BG_SET((gemm_split_glds_kernel<MODE_CONV3S, 2, 2>), 2 * 256 * 2 * GBK * 2);
        BG_SET((gemm_split_glds_kernel<MODE_PLAIN, 4, 3>), 3 * 384 * 2 * GBK * 2 + 14336);
        BG_SET((gemm_split_glds_kernel<MODE_CONV3, 4, 3>), 3 * 384 * 2 * GBK * 2);
        BG_SET((gemm_split_glds_kernel<MODE_CONV3S, 4, 3>), 3 * 384 * 2 * GBK * 2);
        BG_SET((gemm_split_glds_kernel<MODE_PLAIN, 2, 2, true>), 2 * 256 * 2 * GBK * 2 + 14336);
        BG_SET((gemm_split_glds_kernel<MODE_CONV3, 2, 2, true>), 2 * 256 * 2 * GBK * 2);
        BG_SET((gemm_split_glds_kernel<MODE_CONV3S, 2, 2, true>), 2 * 256 * 2 * GBK * 2);
        BG_SET((gemm_split_glds_kernel<MODE_PLAIN, 4, 3, true>), 3 * 384 * 2 * GBK * 2 + 14336);
        BG_SET((gemm_split_glds_kernel<MODE_CONV3, 4, 3, true>), 3 * 384 * 2 * GBK * 2);
        BG_SET((gemm_split_glds_kernel<MODE_CONV3S, 4, 3, true>), 3 * 384 * 2 * GBK * 2);
        BG_SET((gemm_split_glds_kernel<MODE_PLAIN, 2, 2, false, true>), 2 * 256 * 2 * GBK * 2 + 14336);
        BG_SET((gemm_split_glds_kernel<MODE_PLAIN, 2, 2, true, true>), 2 * 256 * 2 * GBK * 2 + 14336);
        BG_SET((gemm_split_glds_kernel<MODE_PLAIN, 2, 4, false, false, 1>), 4 * 256 * 2 * GBK * 2 + 14336);
        BG_SET((gemm_split_glds_kernel<MODE_PLAIN, 2, 4, true, false, 1>), 4 * 256 * 2 * GBK * 2 + 14336);
        BG_SET((gemm_split_glds_kernel<MODE_PLAIN, 2, 4, false, true, 1>), 4 * 256 * 2 * GBK * 2 + 14336);
        BG_SET((gemm_split_glds_kernel<MODE_PLAIN, 2, 4, true, true, 1>), 4 * 256 * 2 * GBK * 2 + 14336);
        BG_SET((gemm_split_glds_kernel<MODE_CONV3, 2, 4, false, false, 1>), 4 * 256 * 2 * GBK * 2);
        BG_SET((gemm_split_glds_kernel<MODE_CONV3S, 2, 4, false, false, 1>), 4 * 256 * 2 * GBK * 2);
        BG_SET((gemm_split_glds_kernel<MODE_CONV3, 2, 4, true, false, 1>), 4 * 256 * 2 * GBK * 2);
        BG_SET((gemm_split_glds_kernel<MODE_CONV3S, 2, 4, true, false, 1>), 4 * 256 * 2 * GBK * 2);
        BG_SET((gemm_split_glds_kernel<MODE_PLAIN, 1, 4, false, false, 1>), 4 * 192 * 2 * GBK * 2 + 14336);
        BG_SET((gemm_split_glds_kernel<MODE_PLAIN, 1, 4, true, false, 1>), 4 * 192 * 2 * GBK * 2 + 14336);
        BG_SET((gemm_split_glds_kernel<MODE_PLAIN, 1, 4, false, false, 1, 1>), 4 * 192 * 2 * GBK * 2 + 14336);
        BG_SET((gemm_split_glds_kernel<MODE_PLAIN, 1, 4, true, false, 1, 1>), 4 * 192 * 2 * GBK * 2 + 14336);
#undef BG_SET
        attr_set[dslot].store(true, std::memory_order_release);
    }
    ProfScope prof(g.mode == MODE_CONV3 ? PROF_CONV3 : (wm == 4 ? PROF_GEMM : PROF_GEMM_SMALL), 2.0 * rows * (double)g.N * g.K, stream);
    const bool conv = g.mode == MODE_CONV3;
    // stride 1, padding 1, no fused upsample: the variant with wave-uniform tap displacements (MODE_CONV3S; $BEVGEN_CONV_FAST=0 keeps the general one, for A/B runs)
    static const int conv_fast_env = getenv("BEVGEN_CONV_FAST") ? atoi(getenv("BEVGEN_CONV_FAST")) : 1;
    const bool convs = conv && conv_fast_env && !g.conv_general && !g.conv_up && g.conv_stride == 1 && g.conv_pad == 1 && g.conv_hin == g.conv_h && g.conv_win == g.conv_w;
#define BG_LAUNCH(MODE_, WM_, S_, THREADS)                                                                                           \
    do {                                                                                                                             \
        if (g.b_lo_zero) hipLaunchKernelGGL((gemm_split_glds_kernel<MODE_, WM_, S_, true>), grid, dim3(THREADS), lds, stream, g);    \
        else hipLaunchKernelGGL((gemm_split_glds_kernel<MODE_, WM_, S_, false>), grid, dim3(THREADS), lds, stream, g);               \
    } while (0)
    if (half8) {
        if (g.b_lo_zero) hipLaunchKernelGGL((gemm_split_glds_kernel<MODE_PLAIN, 1, 4, true, false, 1, 1>), grid, dim3(512), lds, stream, g);
        else hipLaunchKernelGGL((gemm_split_glds_kernel<MODE_PLAIN, 1, 4, false, false, 1, 1>), grid, dim3(512), lds, stream, g);
    } else if (half) {
        if (g.b_lo_zero) hipLaunchKernelGGL((gemm_split_glds_kernel<MODE_PLAIN, 1, 4, true, false, 1>), grid, dim3(256), lds, stream, g);
        else hipLaunchKernelGGL((gemm_split_glds_kernel<MODE_PLAIN, 1, 4, false, false, 1>), grid, dim3(256), lds, stream, g);
    } else if (thin && convs) {
        if (g.b_lo_zero) hipLaunchKernelGGL((gemm_split_glds_kernel<MODE_CONV3S, 2, 4, true, false, 1>), grid, dim3(512), lds, stream, g);
        else hipLaunchKernelGGL((gemm_split_glds_kernel<MODE_CONV3S, 2, 4, false, false, 1>), grid, dim3(512), lds, stream, g);
    } else if (thin && conv) {
        if (g.b_lo_zero) hipLaunchKernelGGL((gemm_split_glds_kernel<MODE_CONV3, 2, 4, true, false, 1>), grid, dim3(512), lds, stream, g);
        else hipLaunchKernelGGL((gemm_split_glds_kernel<MODE_CONV3, 2, 4, false, false, 1>), grid, dim3(512), lds, stream, g);
    } else if (thin) {
        if (g.ksplit > 1) {
            if (g.b_lo_zero) hipLaunchKernelGGL((gemm_split_glds_kernel<MODE_PLAIN, 2, 4, true, true, 1>), grid, dim3(512), lds, stream, g);
            else hipLaunchKernelGGL((gemm_split_glds_kernel<MODE_PLAIN, 2, 4, false, true, 1>), grid, dim3(512), lds, stream, g);
        } else {
            if (g.b_lo_zero) hipLaunchKernelGGL((gemm_split_glds_kernel<MODE_PLAIN, 2, 4, true, false, 1>), grid, dim3(512), lds, stream, g);
            else hipLaunchKernelGGL((gemm_split_glds_kernel<MODE_PLAIN, 2, 4, false, false, 1>), grid, dim3(512), lds, stream, g);
        }
    } else if (g.ksplit > 1) {
        if (g.b_lo_zero) hipLaunchKernelGGL((gemm_split_glds_kernel<MODE_PLAIN, 2, 2, true, true>), grid, dim3(256), lds, stream, g);
        else hipLaunchKernelGGL((gemm_split_glds_kernel<MODE_PLAIN, 2, 2, false, true>), grid, dim3(256), lds, stream, g);
    } else if (wm == 2) {
        if (convs) BG_LAUNCH(MODE_CONV3S, 2, 2, 256);
        else if (conv) BG_LAUNCH(MODE_CONV3, 2, 2, 256);
        else BG_LAUNCH(MODE_PLAIN, 2, 2, 256);
    } else {
        if (convs) BG_LAUNCH(MODE_CONV3S, 4, 3, 512);
        else if (conv) BG_LAUNCH(MODE_CONV3, 4, 3, 512);
        else BG_LAUNCH(MODE_PLAIN, 4, 3, 512);
    }
#undef BG_LAUNCH
    LAUNCH_CHECK();
    if (g.ksplit > 1) launch_splitk_reduce(g, g.kpart, g.ksplit, stream);
}

}  // namespace bevgen
