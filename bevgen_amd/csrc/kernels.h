// Kernel launch interfaces of libbevgen_hip (internal; the public C-ABI is include/bevgen_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bevgen {

// ---------------------------------------------------------------- gemm.hip
enum { MODE_PLAIN = 0, MODE_CONV3 = 1, MODE_CONV3S = 2 };   // (MODE_CONV3S: kernel-template value only - launch_gemm_split_glds picks it for stride-1 / pad-1 / no-upsample convolutions; callers pass MODE_CONV3)
enum { ACT_NONE = 0, ACT_GELU = 1 };
enum { EPI_PLAIN = 0, EPI_MUSE_Q = 1, EPI_GEGLU = 2, EPI_MUSE_KV = 3, EPI_MUSE_QKV = 4 };

struct GemmArgs {
    const float* A = nullptr;  // [M,K] row-major (lda)   | MODE_CONV3: NHWC input [n, Hin, Win, Cin]
    const float* B = nullptr;  // [N,K] row-major (ldb)   | MODE_CONV3: weights [Cout][3][3][Cin]
    float* C = nullptr;        // [M,N] row-major (ldc)
    const float* R = nullptr;  // optional residual [M,N] (ldr), added after the activation
    const float* bias_n = nullptr;  // optional [N]
    const float* bias_m = nullptr;  // optional [M]
    int M = 0, N = 0, K = 0;
    int lda = 0, ldb = 0, ldc = 0, ldr = 0;
    int batch = 1;
    long strideA = 0, strideB = 0, strideC = 0, strideR = 0;
    float alpha = 1.f;
    int act = ACT_NONE;
    int mode = MODE_PLAIN;
    // MODE_CONV3: output spatial size (conv_h x conv_w), input channels, fused nearest-2x upsample of the input
    int conv_h = 0, conv_w = 0, conv_cin = 0, conv_up = 0;
    // general form: stored input is conv_hin x conv_win, tap coordinate = out * conv_stride + k - conv_pad (0 = derive the 'same' defaults:
    // stride 1, pad 1, hin/win = output size or half of it with conv_up).  Downsample (stage1/model.py:56-75): stride 2, pad 0, hin = 2*conv_h.
    int conv_hin = 0, conv_win = 0, conv_stride = 0, conv_pad = -1;
    // split-precision path (gemm_split.hip): the (hi, lo) f16 planes of B in the interleaved layout [N][ldb/32][2][32] (B_lo = B_hi + 32);
    // null = exact fp32 MFMA path
    const uint16_t* B_hi = nullptr;
    const uint16_t* B_lo = nullptr;
    bool b_lo_zero = false;   // B's low plane is all zeros (f16-rounded weight): the LDS-DMA kernel skips the b_lo a_hi product
    // LDS-DMA path (gemm_split_glds.hip): A also arrives as interleaved (hi, lo) f16 planes ([M][lda/32][2][32] / NHWC pixels x channels),
    // written by the producing kernel; A_lo = A_hi + 32
    const uint16_t* A_hi = nullptr;
    const uint16_t* A_lo = nullptr;
    const void* zero_page = nullptr;  // filled in by the launcher
    // fused epilogues of the LDS-DMA kernel: EPI_MUSE_Q writes l2norm(8 x) * epi_scale[d] per head as (hi, lo) f16 planes [B, epi_heads, epi_rows, 64]
    // (rows of the GEMM = B * epi_rows tokens) instead of C
    // EPI_GEGLU (Route M feed-forward, muse_net:71-76): the weight rows are pre-ordered [a(32) | gate(32)] per wave (launch_geglu_weight_order), the epilogue
    // writes gate * gelu(a) - HALF the columns (N/2, ldc) - instead of the raw projection
    int epi = 0;
    const float* epi_scale = nullptr;
    void* epi_hi = nullptr;
    void* epi_lo = nullptr;
    int epi_rows = 0, epi_heads = 0;
    // EPI_MUSE_KV (Route M self-attention, to_kv projection [rows, 2 H 64] = k | v): k columns leave as l2norm(k) * epi_scale planes [B, H, epi_ld, 64] at key row
    // 1 + token (row 0 = the learned null key), v columns as the TRANSPOSED planes [B, H, 64, epi_ld] the attention kernel reads (epi_hi2 / epi_lo2); the row of
    // token 0 also writes the prepared null key / value (epi_aux: [k_hi | k_lo | v_hi | v_lo][H][64] halves)
    void* epi_hi2 = nullptr;
    void* epi_lo2 = nullptr;
    const void* epi_aux = nullptr;
    int epi_ld = 0;
    float epi_post = 1.f;             // EPI_MUSE_Q: extra factor on the prepared query (the attention kernel's score scale, folded in here)
    // MODE_CONV3 on the LDS-DMA kernel, plain epilogue: the GroupNorm(32) statistics of the OUTPUT tensor leave the epilogue as partial sums - per 32 output rows (pixels)
    // and 4 output channels one (sum, sum of squares) pair, gn_part [M / 32][N / 4][2] fp32 - so that the consumer's GroupNorm needs no statistics pass over the tensor
    // (launch_groupnorm_stats_from_partials).  Needs M % 256 == 0 (a row tile never straddles two images: hw % 256 == 0), N % 128 == 0, ldc == N
    float* gn_part = nullptr;
    // EPI_MUSE_QKV (Route M self-attention): to_q and to_kv as ONE projection [rows, 3 H 64] = q | k | v over the concatenated weight - the LayerNorm planes are read once
    // instead of twice and a launch disappears.  Column tiles below H 64 take the EPI_MUSE_Q epilogue with (epi_qh, epi_ql, epi_qscale), the rest the EPI_MUSE_KV one
    // with the fields above (columns counted from H 64)
    void* epi_qh = nullptr;
    void* epi_ql = nullptr;
    const float* epi_qscale = nullptr;
    // LDS-DMA path, small-M problems (low-latency B = 1 scenes): the k range is cut into `ksplit` slices on gridDim.z, every slice leaves its raw fp32 tile sums in
    // kpart [ksplit][M][N] and splitk_reduce adds them in slice order (deterministic) before alpha / bias / activation / residual.  Plain epilogue only.
    int ksplit = 1;
    float* kpart = nullptr;
    int a_bytes = 0;                  // MODE_CONV3: size of the activation plane image (buffer-resource bound), filled in by the launcher
    bool conv_general = false;        // MODE_CONV3 on the LDS-DMA kernel: keep the general variant where the stride-1 one would be picked (operator tests)
    int tile_band = 0;                // tile-order band height (0 = row-major); filled in by the launcher
    int m_base = 0;                   // first row of this launch (the launcher cuts a problem whose last round of tiles would be nearly empty into two row ranges; M stays the END row)
    bool no_row_split = false;        // launcher-internal
    int force_wm = 0;                 // launcher-internal: 4 = keep 256-row blocks (the whole-rounds part of a row-split launch)
    unsigned* status = nullptr;       // the context's device status word (common.h BG_ST_*), filled in by the launchers: an operand outside the f16 range raises BG_ST_F16_RANGE
    bool row_major_epi = false;       // launcher-internal: the plain epilogue of the throughput instantiation stores row-major through LDS (see gemm_tile)
    bool r_prefetch = false;          // launcher-internal (LDS-DMA kernel, 64 x 64 wave patches): the tile's residual lines are requested during the last k-tiles (see gemm_tile)
    // LayerNorm folded across two LDS-DMA GEMMs (Route M, muse_net:62-69: gamma-only LayerNorm, beta is a zero buffer), no pass over the activation in between:
    //   PRODUCER (ln_out_planes != null; EPI_GEGLU, or the plain epilogue): the output rows leave (also / instead of C) as the RAW (hi, lo) planes [M][ln_out_ld / 32][hi | lo]
    //     the consumer reads, plus per (row, 32 output columns) the pair (sum, sum of squares) in ln_out_stats [ln_out_ld / 32][ln_rows][2] - fp32 over 32 values
    //   launch_ln_stats_finalize: one thread per row adds the row's pairs in fp64, in group order -> (mean, rstd) [rows][2] (reads 8 bytes per 32 elements of the activation)
    //   CONSUMER (ln_in_stats != null = that (mean, rstd) array; plain / EPI_MUSE_Q / _KV / _QKV / EPI_GEGLU): B holds W o gamma (columns scaled at finalize), A the raw
    //     planes;  LN(x) W^T = rstd (x (W o gamma)^T - mean cs),  cs[n] = sum_k gamma_k W[n][k] (ln_in_cs), is applied to the tile sums before the rest of the epilogue
    void* ln_out_planes = nullptr;
    float* ln_out_stats = nullptr;
    int ln_out_ld = 0;
    const float* ln_in_stats = nullptr;
    const float* ln_in_gsums = nullptr;   // alternative to ln_in_stats: the producer's group sums [ln_in_groups][ln_rows][2], merged by every workgroup itself (small batches)
    int ln_in_groups = 0, ln_in_count = 0;
    float ln_eps = 1e-5f;
    const float* ln_in_cs = nullptr;
    // stream-K launch (gemm_split_glds_sk_kernel): workspace = 1024 flag words (zeroed by the caller once per workspace) + one partial-tile slot of 128 KiB per workgroup
    // (gemm_sk_ws_bytes); sk_epoch = a value no earlier launch on this workspace used (the flags are never reset: a flag equal to the epoch = "published in THIS launch")
    void* sk_ws = nullptr;
    unsigned sk_epoch = 0;
    int sk_tiles = 0;                 // filled in by the launcher
    bool sk_force = false;            // take the stream-K form whatever gemm_sk_pays says (operator tests)
    int ln_rows = 0;                  // row stride of the producer's statistics array (the whole problem's rows: a row-split launch keeps indexing by absolute row)
};
// (sum, sum of squares) per (32 columns, row) [groups][rows][2] -> (mean, rstd) per row [rows][2]; `count` real columns (the zero padding of a padded row adds nothing to either sum)
void launch_ln_stats_finalize(const float* group_sums, float* row_stats, int rows, int groups, int count, float eps, hipStream_t s);
// W [N, K] -> Wg[n][k] = W[n][k] * gamma[k] (k < Kg, else 0 / W's own padding) and cs[n] = sum_k Wg[n][k] (fp64 accumulation): the consumer-side constants of a folded LayerNorm
void launch_ln_fold_weight(const float* W, const float* gamma, float* Wg, float* cs, int N, int K, int Kg, hipStream_t s);
void launch_gemm(const GemmArgs& g, hipStream_t stream);
void launch_gemm_split_glds(const GemmArgs& g, hipStream_t stream);
bool xcd_placement_verified();                            // decode_fused.hip: workgroup i of a launch runs on XCD i % 8 on the current device (probed once)
size_t gemm_sk_ws_bytes();                                // workspace of a stream-K launch (GemmArgs::sk_ws)
bool gemm_sk_pays(long rows, int N, int K);               // the launcher's rule: does the 256-row tiling leave enough of its last round empty for the stream-K form to win

// Split-precision (3x f16 MFMA, fp32-class accuracy) variant and its weight preparation
struct SplitPlanes { const uint16_t* hi; const uint16_t* lo; bool lo_zero = false; /* the matrix was rounded to f16: its low plane is all zeros */ };
void launch_gemm_split(const GemmArgs& g, hipStream_t stream);
void launch_split_weight(const float* w, void* planes /* 2n halves, interleaved (gemm_split.hip) */, long n, hipStream_t s);
// Table of pre-split weights of the context whose call is executing (null = exact fp32 everywhere): launch_gemm consults it by B pointer.
void split_registry_set(const void* table /* const std::unordered_map<const float*, SplitPlanes>* */);

// Skinny GEMM for decode steps (M <= 64 rows, weight-streaming bound): C[M,N] = A[M,K] B[N,K]^T + bias, act, residual
int gemm_skinny_ksplit(int M, int N, int K);
size_t gemm_skinny_ws_bytes(int M, int N, int K);
void launch_gemm_skinny_ws(const GemmArgs& g, float* ws, hipStream_t stream);  // caller-owned workspace (model path)

// ---------------------------------------------------------------- norm.hip
// y[row, :] = LayerNorm(x[row, :]) * gamma (+ beta); rows of width D; ldx/ldy row strides; columns [D, ldy) of y are zero-filled
void launch_layernorm(const float* x, int ldx, const float* gamma, const float* beta, float* y, int ldy, int rows, int D, float eps, hipStream_t s);
// GEGLU + LayerNorm (muse_net:71-88): h[row, 0:F] = a, h[row, F:2F] = gate -> y = LN(gate * gelu(a)) * gamma ; y row stride ldy (zero padded)
void launch_geglu_layernorm(const float* h, int ldh, const float* gamma, float* y, int ldy, int rows, int F, float eps, hipStream_t s);
// Same two, writing the interleaved (hi, lo) f16 plane image [row][ldy/32][2][32] that the LDS-DMA split-precision GEMM reads (ldy % 32 == 0)
void launch_layernorm_planes(const float* x, int ldx, const float* gamma, const float* beta, void* planes, int ldy, int rows, int D, float eps, hipStream_t s);
void launch_geglu_layernorm_planes(const float* h, int ldh, const float* gamma, void* planes, int ldy, int rows, int F, float eps, hipStream_t s);
// GroupNorm(32 groups, eps) statistics over NHWC [n, hw, C] -> stats[n*32*2] = (mean, rstd)
size_t groupnorm_ws_bytes(int n, int hw);
// decoder tail: GroupNorm(32)-apply + swish + 3x3 convolution C -> 3 + denormalise + NCHW (fp32 y or uint8 y8) in one kernel (x NHWC fp32, wgt [3][3][3][C])
bool vq_out_conv_supported(int C, int cout);
void launch_vq_out_conv(const float* x, const float* stats, const float* gamma, const float* beta, const float* wgt, const float* bias, const float* mean, const float* stdv, int clamp01,
                        float* y, uint8_t* y8, int n, int H, int W, int C, int cout, hipStream_t s);
void launch_groupnorm_stats(const float* x, float* stats, void* ws /*groupnorm_ws_bytes*/, int n, int hw, int C, float eps, hipStream_t s);
// the same statistics from the producing convolution's epilogue partials (GemmArgs::gn_part): no pass over the tensor; fp64 reduction in a fixed order
bool groupnorm_partials_supported(int hw, int C);   // hw % 256 == 0 and whole 4-channel quads per group (C % 128 == 0)
size_t groupnorm_part_floats(int n, int hw, int C);
void launch_groupnorm_stats_from_partials(const float* part, float* stats, int n, int hw, int C, float eps, hipStream_t s);
// y = (x-mean)*rstd*gamma+beta, optionally followed by swish (x*sigmoid(x)); NHWC
void launch_groupnorm_apply(const float* x, const float* stats, const float* gamma, const float* beta, float* y, int n, int hw, int C, int do_swish, hipStream_t s);
void launch_to_planes(const float* x, void* planes, int n, int hw, int C, hipStream_t s);   // fp32 NHWC -> (hi, lo) plane image, values unchanged (C / 4 a power of two)
void launch_groupnorm_apply_planes(const float* x, const float* stats, const float* gamma, const float* beta, void* planes, int n, int hw, int C, int do_swish, hipStream_t s);

// ---------------------------------------------------------------- attention.hip
// Flash attention, fp32 MFMA, head dim 64:  O = softmax(scale * Q K^T + bias) V
//   Q [B,H,Nq,64] (row stride 64), K/V [B,H,Nk_pad,64] with Nk_pad % 32 == 0 (rows >= Nk zero),
//   bias [*, ldb] rows indexed by query, cols by key; entries for key >= Nk must be <= -1e30 (mask baked in);
//   bias may be null only if Nk % 32 == 0.  bias_head_stride = elements between heads (0 = shared by all heads).
//   O (+ optional residual R, same layout) is addressed through o_*stride: [B,Nq,H*64] -> (Nq*H*64, H*64, 64); [B,H,Nq,64] -> (H*Nq*64, 64, Nq*64).
struct AttnArgs {
    const float* Q; const float* K; const float* V; const float* bias; const float* R; float* O;
    int B, H, Nq, Nk_pad;
    long q_bstride, q_hstride;    // elements between batches / heads of Q
    long kv_bstride, kv_hstride;  // same for K and V
    int ldbias; long bias_head_stride;
    float scale;
    long o_bstride, o_qstride, o_hstride;  // O (and R) element index = b*o_bstride + q*o_qstride + head*o_hstride + d
    int kv_group = 1;             // consecutive groups of kv_group batches read the K / V of batch b / kv_group (samples of one BEV layout share the condition's cross-attention K / V)
    // block-sparse layouts (Route A prefill, the SparseSelfAttention operator): per (head, 128-row query block) the ascending list of 32-key tiles in which ANY of the
    // block's rows sees a key - count first, [heads or 1][cdiv(Nq, 128)][tiles_ld] (launch_build_attn_tiles from the masked bias); tiles outside the list are never
    // loaded or multiplied (sparse_self_attention.py:63-85 computes only the nonzero layout blocks).  null = every tile
    const uint16_t* tiles = nullptr; long tiles_head_stride = 0; int tiles_ld = 0;
};
void launch_attention(const AttnArgs& a, hipStream_t s);
size_t attn_tiles_elems(int heads, int Nq, int Nk_pad);
void launch_build_attn_tiles(const float* masked_bias, long bias_head_stride, int ldbias, int heads, int Nq, int Nk_pad, uint16_t* tiles, hipStream_t s);
void launch_splitk_reduce(const GemmArgs& g, const float* partial, int ksplit, hipStream_t s);   // gemm_skinny.hip: C = act(alpha * sum_k partial[k] + bias) + R

// Split-precision flash attention (attention_split.hip): operands pre-split into (hi, lo) f16 planes by the preparation kernels.
//   The softmax is evaluated in the base-2 domain: `bias` must arrive PRE-MULTIPLIED by log2(e) (kLog2e below) and the score scale (x log2 e) must
//   already be folded into the Q planes (EPI_MUSE_Q epilogue / muse_q_prep_split's `post` factor); `scale` is ignored.
//   Qh/Ql [B,H,Nq,64], Kh/Kl [B,H,Nk_pad,64], VTh/VTl [B,H,64,Nk_pad] (V transposed); bias / output conventions as AttnArgs.
constexpr float kLog2e = 1.44269504088896340736f;
struct AttnSplitArgs {
    const _Float16 *Qh, *Ql, *Kh, *Kl, *VTh, *VTl;
    const float* bias;   // row-major [Nq, ldbias]: read only by the round-1 kernel of tools/attn_lab
    float* O;
    int B, H, Nq, Nk_pad;
    int ldbias; long bias_head_stride;
    float scale;
    long o_bstride, o_qstride, o_hstride;
    _Float16* Op;   // non-null: write the output as interleaved (hi, lo) planes of the [B*Nq, H*64] matrix instead of fp32 O
    const float* bias_pk = nullptr;   // packed bias image (launch_pack_attn_bias) of the [Nq, Nk_pad] bias, or null; bias_head_stride applies to it
    int bias_pk_tile_step = 0;        // set by the launcher
    long bias_pk_qb_stride = 0;
    int kv_group = 1;                 // as AttnArgs::kv_group
    // key split (low-latency path: one scene gives cdiv(Nq, 256) * H = 96 workgroups for 256 CUs): the key tiles are cut into `ksplit` ranges, one workgroup per
    // (query block, head, batch, range); each writes its unnormalised output row, running maximum and row sum to `kws` and a combine kernel merges the ranges
    int ksplit = 1;
    float* kws = nullptr;             // attn_split_ws_floats(B, H, Nq, ksplit) floats
    unsigned* status = nullptr;       // the context's device status word, filled in by the launcher (an output plane value outside the f16 range: BG_ST_F16_RANGE)
};
constexpr int kAttnPartPitch = 68;   // floats per partial row of the key-split Route-M attention: 64 output columns, running maximum, row sum, 2 of padding (16-byte rows)
inline long attn_split_ws_floats(int B, int H, int Nq, int ksplit) { return (long)ksplit * B * H * Nq * kAttnPartPitch; }
long attn_bias_packed_floats(int Nq, int Nk_pad);
void launch_pack_attn_bias(const float* bias, int ld, int Nq, int Nk_pad, float* out, hipStream_t s);
void launch_attn_split_operands(const float* q, const float* k, const float* v, void* Qh, void* Ql, void* Kh, void* Kl, void* VTh, void* VTl, int B, int H, int Nq, int Nk_pad,
                                float qmul, hipStream_t s);
void launch_attention_split(const AttnSplitArgs& a, hipStream_t s);
void launch_muse_q_prep_split(const float* qraw, const float* q_scale, void* Qh, void* Ql, int B, int H, int Nq, float post, hipStream_t s);
void launch_muse_null_kv_prep(const float* null_kv, const float* k_scale, void* out /* 4*H*64 halves */, int H, hipStream_t s);
void launch_muse_kv_prep_split(const float* kvraw, const float* null_kv, const float* k_scale, void* Kh, void* Kl, void* VTh, void* VTl, int B, int H, int Nk,
                               int Nk_pad, hipStream_t s);

// Route A visibility of key k for query row r of head h:  attention_mask[r][k] != 0  AND  layout[h][r / blk][k / blk] != 0
// (sparse_self_attention.py:63-85: only the nonzero blocks of the layout are ever computed; :153-173: the element mask inside them).
// Kept as its two factors - ONE [L, L] byte plane shared by all layers / heads and the block layout - plus, per block row, the ascending list of the
// 16-key chunks that contain a present block: the decode attention walks that list and never touches the K/V rows of absent blocks (no dense
// [layers][heads][L][L] plane exists any more: 2.15 GB at BASELINE config 4 / density 0.35, now ~31 MB).
struct SparseVis {
    const uint8_t* allowed = nullptr; int ldallowed = 0; long allowed_head_stride = 0;   // [(H)][L][ld] 1 = allowed; null = every (row, key)
    const uint8_t* lay = nullptr; long lay_head_stride = 0; int nb = 0, blk = 1;          // [H or 1][nb][nb] 1 = block present; null = every block
    const uint16_t* chunks = nullptr; long chunks_head_stride = 0; int chunks_ld = 0;     // [H or 1][nb][chunks_ld]: count, then ascending chunk ids; null = walk all chunks
    int has_allowed = 0, has_lay = 0, has_chunks = 0;                                     // filled in by vis_fix (absent tables alias valid memory)
};
// kernels load table entries UNCONDITIONALLY (a predicated load is a serialised round trip, see decode_fused.hip): absent tables point at `any_valid` (>= 4 readable bytes)
inline SparseVis vis_fix(SparseVis v, const void* any_valid) {
    v.has_allowed = v.allowed != nullptr; v.has_lay = v.lay != nullptr; v.has_chunks = v.chunks != nullptr;
    if (!v.allowed) { v.allowed = reinterpret_cast<const uint8_t*>(any_valid); v.ldallowed = 0; v.allowed_head_stride = 0; }
    if (!v.lay) { v.lay = reinterpret_cast<const uint8_t*>(any_valid); v.lay_head_stride = 0; v.nb = 0; v.blk = 1 << 30; }
    if (!v.chunks) { v.chunks = reinterpret_cast<const uint16_t*>(any_valid); v.chunks_head_stride = 0; v.chunks_ld = 0; }
    return v;
}

// Decode attention (Route A, one new query row per sequence): see attention.hip
struct DecodeAttnArgs {
    const float* q = nullptr;        // [B, H*64] this step's query rows (row stride ldq)
    int ldq = 0;
    const void* kcache = nullptr;    // [B, H, Lmax, 64] storage dtype
    const void* vcache = nullptr;
    const float* bias = nullptr;     // [L, ldbias] camera-bias matrix (unscaled; row = n-1 is used) or null
    int ldbias = 0;
    SparseVis vis;                   // which keys the row sees (element mask x block layout); this kernel masks per key, it does not skip
    const float* append_k = nullptr; // this step's key / value rows [B, H*64] (row stride ldq): written into cache row n-1 by the kernel itself
    const float* append_v = nullptr; //   (fused KV append, saves one launch per layer); null = the cache already holds row n-1
    const float* R = nullptr;        // residual [B, H*64] (row stride ldr) or null
    float* O = nullptr;              // [B, H*64] (row stride ldo)
    int ldr = 0, ldo = 0;
    int B = 0, H = 0;
    int n = 0;                       // context length (keys 0..n-1); if d_n != null the length is *d_n + n
    const int* d_n = nullptr;        // device-side step counter (hipGraph replay)
    int n_hint = 0;                  // host's copy of *d_n (profiling only)
    int Lmax = 0;
    float scale = 1.f;               // dh^-0.5, applied to (q.k + bias)
    int kv_dtype = 0;                // 0 = fp32, 1 = fp16 storage (fp32 accumulate)
    int shared_prefix = 0;           // reserved: leading keys shared by groups of `group` consecutive sequences
    int group = 1;
    unsigned* status = nullptr;      // the context's device status word, filled in by the launcher
};
int decode_attention_splits(int B, int H, int n_max);
size_t decode_attention_ws_bytes(int B, int H, int S);
void launch_decode_attention_ws(const DecodeAttnArgs& a, float* ws, int S, hipStream_t s);
void launch_decode_attention_combine(const DecodeAttnArgs& a, const float* ws, int S, hipStream_t s);   // o / l (+ R) from [B][H][S][66] partials

// ---------------------------------------------------------------- decode_fused.hip (Route A decode step, three launches per layer)
// A [M, D] fp32 matrix that may still be "in flight" as split-K partial sums: element (m, c) = base[m*ld + c] + bias[c] + sum_k partial[k*pstride + m*pld + c]
// (partials added in index order: deterministic).  ns = 0 / partial = bias = null -> the plain matrix `base`.
struct RowSrc {
    const float* base = nullptr; int ld = 0;
    const float* partial = nullptr; int ns = 0; long pstride = 0; int pld = 0;
    const float* bias = nullptr;
    int has_bias = 0;                  // filled in by the launchers (absent tables alias `base` so that kernels can load unconditionally)
};
void launch_rowsrc_materialize(const RowSrc& r, float* out, int M, int D, int* counter /* or null: incremented by one */, hipStream_t s);

struct ArAttnFusedArgs {
    RowSrc x;                          // rows entering the layer (before ln1), [B, D]
    const float *ln_w = nullptr, *ln_b = nullptr; float eps = 1e-5f;
    const float* qkv = nullptr;        // non-null: q | k | v rows [B, 3D] already projected by the LN + QKV kernel (decode_path = split); `xn` then holds ln1(x) [B, D]
    const float* xn = nullptr;         //   and x / ln_w / wqkv are not read
    const float* wqkv = nullptr;       // fused [3D, D] (q | k | v), bqkv [3D]
    const void* wqkv_h = nullptr;      // non-null: the same matrix stored as fp16 (decode_weights = f16), read instead of wqkv
    const float* bqkv = nullptr;
    const float *ln_cs = nullptr, *ln_ds = nullptr;   // [3D] each: W gamma and W beta + b of the fused matrix (launch_ar_ln_fold): the fused kernel folds ln1 into its projection
    void* kcache = nullptr;            // this layer's [B, H, Lmax, 64]
    void* vcache = nullptr;
    int kv_dtype = 0;                  // 0 fp32, 1 fp16 storage
    const float* bias = nullptr; int ldbias = 0;                               // camera-bias matrix [L, ldbias] (unscaled) or null
    SparseVis vis;                     // visibility of the keys (element mask x block layout) + the chunk lists the key walk follows
    float* out = nullptr; int ldo = 0; // x2 [B, D] = ln1(x) + attention
    int B = 0, G = 1, H = 0, D = 0, Lmax = 0, Lpad = 0;
    int n = 0; const int* d_n = nullptr; int n_hint = 0;                        // context length incl. the new key = n (+ *d_n)
    int prefix = 0;                    // G > 1: leading keys shared by the G sequences of a group (read from the group's first cache slot)
    float scale = 0.125f;
    long long* trace = nullptr;        // diagnostics: [workgroup][8] device timestamps (100 MHz) at the phase boundaries, or null
    // key split (attention-only kernel, G = 1, few sequences): gridDim.z = ksplit workgroups share one (sequence, head), each walks a contiguous range of the
    // list positions and leaves (max, sum, unnormalised output[64]) in kws [B][H][ksplit][66]; launch_ar_attn_fused then runs the combine kernel (+ residual)
    int ksplit = 1;
    float* kws = nullptr;
    // K/V rows staged in LDS (fused kernel, one sequence per workgroup; dense walk or chunk-list walk): every wave requests the leading whole pipeline steps of ITS OWN
    // share of the key walk - `stage_cap` 1 KiB pieces, K and V of a step together - by LDS-DMA behind the last product of the projection: a deep, register-free
    // prefetch that starts as early as the CU's in-order memory pipeline allows without stalling the projection; the walk reads those steps from LDS and the rest
    // from HBM.  -1 = the launcher's choice ($BEVGEN_KV_STAGE overrides), 0 = off
    int stage_cap = -1;
    int has_bias = 0;                  // filled in by the launcher
    unsigned* status = nullptr;        // the context's device status word, filled in by the launcher (fp16 cache: a key / value outside the fp16 range raises BG_ST_F16_RANGE)
};
bool ar_attn_fused_supported(int B, int G, int D, int H);
// per-row constants of LayerNorm folded into a projection: cs[j] = sum_k W[j][k] gamma[k], ds[j] = sum_k W[j][k] beta[k] + b[j]   (W [N, K] row-major; fp64 accumulation)
void launch_ar_ln_fold(const float* W, const float* b, const float* gamma, const float* beta, float* cs, float* ds, int N, int K, hipStream_t s);
void launch_ar_attn_fused(const ArAttnFusedArgs& a, hipStream_t s);

struct SkinnyFusedArgs {
    const float* A = nullptr; int lda = 0;   // [M, K]
    const RowSrc* a_src = nullptr;           // non-null: A is this row source (previous layer's split-K partials + bias + residual, added while the tile is fetched)
    float* xn_out = nullptr; int ldxn = 0;   // non-null (with LayerNorm): the normalised rows are also written out [M, K] (the residual Block.forward takes from ln1(x))
    const float *ln_w = nullptr, *ln_b = nullptr; float eps = 1e-5f;   // ln_w != null: LayerNorm over K fused in front of the product
    const float *ln_cs = nullptr, *ln_ds = nullptr;   // [N] each, non-null (plain A + LayerNorm): W gamma and W beta + bias (launch_ar_ln_fold) - the LayerNorm is folded into the product, `bias` is not read
    const float* Wp = nullptr;               // [N, K] weights in the packed operand layout (launch_pack_skinny_weight / _f16)
    int w_f16 = 0;                           // the packed image holds fp16 values
    const float* bias = nullptr;       // [N] (ksplit == 1 only)
    float* C = nullptr; int ldc = 0;   // [M, N], or the partial sums [ksplit][M][N] when ksplit > 1
    int M = 0, N = 0, K = 0, ksplit = 0 /* 0 = skinny_fused_ksplit(N, K) */, act = 0;
    long long* trace = nullptr;        // diagnostics, as above
    int has_ln_b = 0;                  // filled in by the launcher
    RowSrc src;                        // filled in by the launcher from a_src
};
// Both MLP projections of a decode layer in ONE launch (decode_fused.hip: ar_mlp_fused_kernel).  One workgroup per 16 hidden columns (4 D / 16 of them: 256 at D = 1024,
// one per CU); workgroup i runs on XCD i % 8 and the 4 D / 128 workgroups of an XCD produce a contiguous D / 2 slice of the hidden row, which is exactly one K slice of the
// down-projection: after an XCD-LOCAL exchange (plain stores, one L2 counter, no agent-scope fence: producers and consumers share the XCD's L2) every workgroup multiplies
// that slice by its 32 output columns of the down matrix, whose weights it requested at kernel start.  Output: 8 partial planes [8][M][D] (one per XCD), summed in order by
// the consumer's row source.  M <= 64 (row chunks of 16), D = 1024, a device with at least 4 D / 16 CUs; the launch must have the GPU to itself (every workgroup of an XCD waits for
// its 31 peers: see the co-residency note in decode_fused.hip).
struct MlpFusedArgs {
    const float* A = nullptr; int lda = 0;            // [M, D] rows entering ln2
    const float* ln_w = nullptr;                       // ln2 gamma [D]
    const float *ln_cs = nullptr, *ln_ds = nullptr;    // [4 D] each: W_up gamma and W_up beta + bias (launch_ar_ln_fold)
    float eps = 1e-5f;
    const float* Wup = nullptr;                        // packed operand image of the up matrix [4 D, D]
    const float* Wdn = nullptr;                        // packed operand image of the down matrix [D, 4 D]
    int w_f16 = 0;
    float* hidden = nullptr;                           // [M][4 D] scratch (the GELU output, exchanged through L2)
    float* C = nullptr;                                // [8][M][D] partial sums of the down-projection
    unsigned* sync = nullptr;                          // mlp_fused_sync_words() words of device memory, zeroed once (barrier state per XCD + placement check)
    unsigned* err = nullptr;                           // the context's host-visible status word (common.h BG_ST_*): barrier timeout, wrong placement, f16 operand range
    int M = 0, D = 0;
    long long* trace = nullptr;
    int acq = 1;                                       // acquire form of the exchange (decode_fused.hip), filled in by the launcher
};
constexpr int MLP_FUSED_PLANES = 8;
size_t mlp_fused_sync_words();
bool mlp_fused_supported(int M, int D, bool w_f16);    // shape + device (CU count) check
void launch_ar_mlp_fused(const MlpFusedArgs& g, hipStream_t s);
size_t skinny_packed_floats(int N, int K);
void launch_pack_skinny_weight(const float* W, float* Wp, int N, int K, hipStream_t s);
void launch_pack_skinny_weight_f16(const float* W, void* Wp /* N16 * K halves */, int N, int K, hipStream_t s);
int skinny_fused_ksplit(int N, int K);
bool skinny_fused_supported(int M, int N, int K, bool ln);
bool skinny_fused_f16_ok(int N, int K, bool ln);   // fp16 weight image: K slice per wave a multiple of 32
size_t ar_attn_fused_lds_bytes(int G, int D, int Lpad);
size_t ar_attn_fused_max_lds();   // LDS bytes a workgroup may allocate on the current device
size_t ar_attn_lds_bytes(int G, int Lpad);   // the attention-only kernel (decode_path = split)
void launch_skinny_fused(const SkinnyFusedArgs& g, hipStream_t s);

// ---------------------------------------------------------------- embed.hip
// Geometric camera embeddings (gpt:336-349 / muse_net:314-327)
//   c_embed[b,c,:] = Wcam[D,4] * E_inv[b,c,:,3];  img[b,c,t,:] = normalize(Wimg * (E_inv [I_inv pix_t; 1]) - c_embed) (+1e-7)
void launch_camera_embed(const float* I_inv, const float* E_inv, const float* plane /*[3,T]*/, const float* Wimg, const float* Wcam,
                         float* img /*[B,C,T,D]*/, float* c_embed /*[B,C,D]*/, int B, int C, int T, int D, hipStream_t s);
// context[b,k,:] = cond_tok[cond_ids[b,k]] + (grid[k,:]*Wbev + bbev - sum_c(bev_cam_pos[c,k,:] + c_embed[b,c,:])) + cond_pos[k,:]
void launch_cond_embed(const int64_t* cond_ids, const float* cond_tok, const float* cond_pos, const float* bev_grid /*[3,K] or null*/,
                       const float* Wbev, const float* bbev, const float* bev_cam_pos /*[C,K,D]*/, const float* c_embed /*[B,C,D]*/,
                       float* out, int B, int C, int K, int D, int cond_vocab, hipStream_t s);
// x[b,n,:] = (tok_emb[ids[b,n]] + img[b,n,:]) + pos[n,:]        (muse_net:309-331)
void launch_token_embed(const int64_t* ids, const float* tok_emb, const float* img /*or null*/, const float* pos, float* x,
                        int B, int N, int D, int vocab_rows, hipStream_t s);

// Route M q/k/v preparation (muse_net:132-146): l2norm + per-dim scale, null-kv prepended, head-major layout
//   qraw [B*Nq, H*64] -> Q [B,H,Nq,64];  kvraw [B*Nk, 2*H*64] (k first, v second) -> K,V [B,H,Nk_pad,64] rows 1..Nk (row 0 = null kv)
void launch_muse_q_prep(const float* qraw, const float* q_scale, float* Q, int B, int H, int Nq, hipStream_t s);
void launch_muse_kv_prep(const float* kvraw, const float* null_kv /*[2,H,1,64]*/, const float* k_scale, float* K, float* V,
                         int B, int H, int Nk, int Nk_pad, hipStream_t s);
// Route A: split fused qkv rows [rows, 3*H*64] (q|k|v) -> Q [B,H,n,64] (optional) and cache rows [B,H,Lmax,64] at position pos0..pos0+n
void launch_ar_qkv_scatter(const float* qkv, float* Q /*[B,H,n,64] or null*/, void* kcache, void* vcache, int kv_dtype,
                           int B, int H, int n, int pos0, int Lmax, hipStream_t s);
// Route A decode: pos0 read from a device counter (graph-replayable)
void launch_ar_kv_append(const float* qkv, void* kcache, void* vcache, int kv_dtype, int B, int H, int pos0, const int* d_pos, int Lmax, hipStream_t s);

// ---------------------------------------------------------------- sampler.hip
// MaskGit re-masking (muse_net:569-574): the n_mask highest scores of each row (ties: lower index first) get mask_id; init ids re-imposed
void launch_remask(int64_t* ids, const float* scores, const int64_t* init_ids /*or null*/, int rows, int T, int n_mask, int64_t mask_id, hipStream_t s);
// MaskGit token pick (muse_net:587-599): top-k filter, /max(temp,1e-10), + gumbel(u), argmax; only masked positions are overwritten
void launch_maskgit_pick(int64_t* ids, const float* logits, int ldl, const float* gumbel_u /*or null*/, int rows, int V, int k, float temperature,
                         int64_t mask_id, hipStream_t s, unsigned long long seed = 0 /* != 0 and gumbel_u null: uniforms from Philox(seed, iter) */, unsigned iter = 0,
                         float* conf_scores = nullptr, int conf_mode = 0 /* scores without a token critic (muse_net:611-622): 1 = masked positions only, 2 = everywhere */);
// Self-critic scores (muse_net:392-396, 602-611): scores = embed . w + b + ((u - 0.5) * noise_scale) * frac
void launch_critic_scores(const float* embed, int lde, const float* w, const float* b, const float* u /*or null*/, float noise_scale, float frac,
                          float* scores, int rows, int D, hipStream_t s, unsigned long long seed = 0, unsigned iter = 0);
void launch_philox_fill(float* out, long n, unsigned long long seed, unsigned iter, unsigned stream, int V /* stream 0: vocabulary size (row layout) */, hipStream_t s);
// Route A token pick (ar_lm:204-219): logits/temperature, top-k (ties kept), softmax, argmax or inverse-CDF draw with explicit u
//   forced [steps, rows] (or null): entries >= 0 are emitted instead of a drawn token (partial decoding, ar_lm:161-165,181-182)
// `tail` (optional): what the decode step does with the token next, done by the same launch - out_all[row, fwd_idx[step]] = token (ar_lm:219) and the new row's embedding
// x[row, :] = x_tok_emb[token] + img_embed[row, fwd_idx[step]] + x_pos_emb[fwd_idx[step]] (gpt:331-365 for the one new row): two launches per step less
struct ArPickTail {
    int64_t* out_all = nullptr; const int64_t* fwd_idx = nullptr; int N = 0;
    const float *tok_emb = nullptr, *img_embed = nullptr, *pos_emb = nullptr; float* x = nullptr; int C = 0, T = 0, D = 0, vocab_rows = 0;
};
void launch_ar_pick(const float* logits, int ldl, const float* u /*[steps, rows] or null*/, const int* d_step /*or null: u is this step's row*/, const int64_t* forced,
                    int64_t* out, int rows, int V, int top_k, float temperature, hipStream_t s, const ArPickTail* tail = nullptr);

// ---------------------------------------------------------------- vq.hip
void launch_codebook_gather(const int64_t* ids, const float* codebook, float* out, int rows, int dim, int n_embed, hipStream_t s);
// NHWC [n,hw,C] -> NCHW [n,C,hw] with optional per-channel x*std+mean and clamp to [0,1] (bev_utils/util.py:97-118)
void launch_nhwc_to_nchw(const float* x, float* y, int n, int hw, int C, int ldc, const float* mean, const float* stdv, int clamp01, hipStream_t s,
                         uint8_t* y8 = nullptr /* non-null: write round(v*255) as uint8 here instead of fp32 y */);
void launch_nchw_to_nhwc(const float* x, float* y, int n, int hw, int C, hipStream_t s);
void launch_row_softmax(float* x, int rows, int cols, float scale, hipStream_t s, int ld = 0 /* row stride (0 = cols); columns [cols, ld) are zero-filled */);

// misc
void launch_fill(float* p, long n, float v, hipStream_t s);
void launch_add(const float* a, const float* b, float* c, long n, hipStream_t s);
void launch_scale(float* p, long n, float v, hipStream_t s);

}  // namespace bevgen
