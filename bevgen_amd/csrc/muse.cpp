// Route M: MaskGit bidirectional decoder.
//   TransformerMultiView.forward      stage2/muse_maskgit_pytorch.py:283-371
//   TransformerBlocks / Attention     :171-202 / :90-169        FeedForward :78-88
//   MaskGit.generate                  :511-627                   SelfCritic  :388-396
//
// What is hoisted out of the 18-iteration loop (all of it is arithmetic the reference repeats identically 72 times per call):
//   * geometric image embedding + condition (BEV) embedding          (depend only on cameras / cond ids)
//   * cross-attention K/V of every layer (to_kv(context) + l2norm)    (context is never updated)
//   * the attention-bias matrices                                     (built once in bevgen_finalize)
// and what is dropped: the classifier-free-guidance "null" forwards (bit-identical to the conditional ones in eval mode,
// muse_net:352) and the critic forward after the last iteration (its scores are never read).
#include "model.h"

namespace bevgen {

namespace {

struct MuseWs {
    int B = 0;
    int S = 1;   // samples per BEV layout: consecutive groups of S scenes share their condition, whose cross-attention K / V exist once per layout

    long rows = 0;
    float *img = nullptr, *c_embed = nullptr, *context = nullptr;
    std::vector<float*> crossK, crossV;
    float *x = nullptr, *xn = nullptr, *qraw = nullptr, *kvraw = nullptr, *Q = nullptr, *Ks = nullptr, *Vs = nullptr, *att = nullptr, *h = nullptr, *g = nullptr;
    float* attn_ws = nullptr; int attn_ks = 1;   // key-split self-attention of the low-latency path (pick_attn_ksplit): partial rows of the key ranges
    float* kpart = nullptr;   // split-K partial tiles of the narrow (N = D) projections when the batch is too small to fill the chip (low-latency path)
    float *stats_d = nullptr, *stats_f = nullptr;   // folded LayerNorms: per-(32 columns, row) (sum, sum of squares) of the residual rows [D / 32][rows][2] / of the GEGLU rows [Fpad / 32][rows][2]
    float* rowstat = nullptr;                       // ... merged to (mean, rstd) per row [rows][2] by launch_ln_stats_finalize
    void* sk_ws = nullptr;                          // stream-K workspace of the projections (small batches; flags zeroed in muse_prepare), null = never stream-K
    unsigned sk_epoch = 0;                          // bumped per projection launch
};

// LayerNorm folded across the GEMMs around it (GemmArgs::ln_*; needs the fold constants of bevgen_finalize: split-precision mode with fp32 weights).
//   0  the LayerNorm kernels of rounds 1-5 everywhere
//   1  (default) the feed-forward's INNER LayerNorm (over F = 2730 columns, the widest pass: 3.3 % of the sixteen-scene step) disappears: the GEGLU epilogue writes raw
//      planes + per-32-column row sums INSTEAD of its fp32 result (no extra bytes), a one-thread-per-row kernel merges the sums (8 bytes read per 32 elements), the
//      down-projection multiplies by W4 o gamma and applies rstd (acc - mean cs)
//   2  all four LayerNorms of a layer (the residual-stream projections then write planes + sums BESIDES their fp32 row)
// Measured, same box, scenes/s at 16 / 2 / 1 scenes per call (profiles/r06_ab_ln_fold.txt): level 0 10.29 / 7.88 / 6.04, level 1 **10.53 / 7.98 / 6.06**, level 2 10.10 / 7.85 /
// 5.95 - the extra plane stores of the residual-stream epilogues cost more than the LayerNorm passes they replace (round 3 found the same with another design), at
// every batch size; level 2 stays as a tested switch.  $BEVGEN_LN_FOLD pins the level (A/B runs, tests).
int ln_fold_level(const Ctx& c, long rows) {
    const char* e = getenv("BEVGEN_LN_FOLD");   // (read per forward, not cached: the tests switch it inside one process)
    const int env = e ? atoi(e) : 1;
    (void)rows;
    if (c.muse.empty() || !c.muse[0].fold_w4) return 0;
    return std::max(0, std::min(env, 2));
}

// Low-latency path (one or two scenes per call, scripts/interactive_editing.py:273-277): a [rows, D] x [D, D] projection is only cdiv(rows, 128) * D / 128 tiles -
// 96 workgroups at one six-view scene, on 256 CUs.  Its k range is cut into slices until the grid covers the chip; the slices' partial tiles are added in a fixed
// order (tokens stay identical run to run, and identical to the unsplit path up to fp32 summation order - checked against the B = 16 path in the tests).
// How many slices (measured at one six-view scene, rows = 1536, tools/gemm_small_probe.py + profiles/r03_gemm_small_*.txt): with the eight-wave small-problem block of
// gemm_split_glds.hip a [1536, 1024] x [1024, K] projection costs 7.6 us of fixed time (dispatch, pipeline fill from cold operands, store tail) + 0.43 us per k-tile
// on 96 CUs.  A second slice halves the loop but adds a 6.4 us reduce launch and gives up the fused epilogues (q preparation): a wash at K = 1024 (21.4 vs 20.9 us),
// a clear gain for a long k loop (K = 5504: 82 -> 51 us).  So: two slices from K = 2048 up, none below - and none where the 128-row grid covers at most half of the CUs:
// there the launcher's 64-row blocks double the grid without a reduce launch (one-scene down-projection, K = 2752: 37.9 + 7.1 -> 40 us; step 169.9 -> 167.9 ms).  (Round-3 history: with the four-wave
// block - 12 us + 0.62 us per k-tile - two slices everywhere were the optimum, step 257 -> 241 ms.)
constexpr int KSPLIT_MAX = 6;
constexpr long kSkMaxRows = 8192;   // batches up to this many token rows carry a stream-K workspace (gemm_sk_pays decides per projection)
int ksplit_env() { static const int v = getenv("BEVGEN_KSPLIT") ? atoi(getenv("BEVGEN_KSPLIT")) : 0; return v; }
int pick_ksplit(long rows, int N, int K) {
    if ((long)cdiv(rows, 256) * cdiv(N, 128) >= 256) return 1;          // the 256-row tiling already fills the chip
    const long tiles = (long)cdiv(rows, 128) * cdiv(N, 128);
    if (tiles >= 160) return 1;
    int s = (K >= 2048 && tiles > 128) ? 2 : 1;   // (<= 128 tiles: the launcher's 64-row blocks already double the grid, without a reduce: 45 -> 40 us at K = 2752)
    if (ksplit_env() > 0) s = std::min(KSPLIT_MAX, ksplit_env());
    while (s > 1 && K / 32 < 2 * s) --s;
    return s;
}

// Key ranges of the self-attention (same low-latency path): one scene is cdiv(1536, 256) * 16 heads = 96 workgroups on 256 CUs, each walking all 49 key tiles;
// with the keys cut in two, 192 workgroups walk half of them and a combine kernel merges the pairs.  Only when the ranges stay long (>= 8 tiles each) and the
// grid leaves at least half of the CUs idle.  $BEVGEN_ATTN_KSPLIT pins it (A/B runs, tests).
int pick_attn_ksplit(long blocks, int ntiles) {
    static const int env = getenv("BEVGEN_ATTN_KSPLIT") ? atoi(getenv("BEVGEN_ATTN_KSPLIT")) : 0;
    if (env > 0) return std::max(1, std::min(std::min(env, 8), ntiles));
    if (blocks * 2 > 256) return 1;
    const int ks = std::min(std::min(4, (int)(256 / blocks)), ntiles / 8);
    return std::max(ks, 1);
}

size_t muse_ws_bytes(const Ctx& c, int B) {
    const size_t rows = (size_t)B * c.N;
    const size_t crows = (size_t)B * c.K;
    size_t f = 0;
    f += rows * c.D * 2;                 // img, x
    f += (size_t)B * c.cfg.num_cams * c.D + crows * c.D;
    f += (size_t)c.cfg.num_layers * 2 * B * c.H * c.NkC_pad * 64 + (size_t)B * c.K * 2 + (size_t)B * c.cfg.num_cams * c.D;
    f += rows * c.D * 3;                 // xn, qraw, att
    f += std::max(rows, crows) * 2 * c.D; // kvraw
    f += rows * c.D;                     // Q
    f += (size_t)2 * B * c.H * c.NkS_pad * 64;
    f += rows * 2 * c.F + rows * c.Fpad;
    f += rows * (c.V + 1);               // logits, scores (generate)
    if (std::max(pick_ksplit((long)rows, c.D, c.D), pick_ksplit((long)rows, c.D, c.Fpad)) > 1) f += (size_t)KSPLIT_MAX * rows * c.D;
    {
        const int ks = pick_attn_ksplit((long)cdiv(c.N, 256) * c.H * B, c.NkS_pad / 32);
        if (ks > 1) f += (size_t)attn_split_ws_floats(B, c.H, c.N, ks);
    }   // split-K partial tiles (small batches only)
    f += rows * (size_t)(2 * (c.D / 32) + 2 * (c.Fpad / 32) + 4);   // folded-LayerNorm row statistics
    return f * sizeof(float) + (64 + 4 * c.cfg.num_layers) * 256 + (rows <= kSkMaxRows ? gemm_sk_ws_bytes() + 256 : 0);
}

void gemm(const float* A, int lda, const float* W, int ldb, float* C, int ldc, int M, int N, int K, const float* R, int ldr, hipStream_t s) {
    GemmArgs g;
    g.A = A; g.B = W; g.C = C; g.R = R;
    g.M = M; g.N = N; g.K = K;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ldr = ldr;
    launch_gemm(g, s);
}

// same with A given as interleaved (hi, lo) f16 planes [M][lda/32][2][32] (split-precision mode: the producer kernel wrote them)
// folded-LayerNorm roles of a projection (GemmArgs::ln_*): `in` = it consumes raw planes + the statistics of `in_groups` 32-column groups over `in_count` real columns
// (B must then be the W o gamma matrix, cs its row sums); `out_planes` = it produces raw planes + statistics of its own output rows
struct LnFold {
    const float* in_stats = nullptr; const float* in_cs = nullptr;   // consumer: per-row (mean, rstd) [rows][2] and the column sums of W o gamma
    const float* in_gsums = nullptr; int in_groups = 0, in_count = 0; // ... or (small batches) the producer's group sums, merged by the consuming workgroups themselves
    void* out_planes = nullptr; float* out_stats = nullptr; int out_ld = 0;
};
void set_sk(GemmArgs& g, MuseWs* w);
void set_fold(GemmArgs& g, const LnFold* f, int rows) {
    if (!f) return;
    g.ln_in_stats = f->in_stats; g.ln_in_cs = f->in_cs;
    g.ln_in_gsums = f->in_gsums; g.ln_in_groups = f->in_groups; g.ln_in_count = f->in_count;
    g.ln_out_planes = f->out_planes; g.ln_out_stats = f->out_stats; g.ln_out_ld = f->out_ld;
    g.ln_rows = rows;
}

void gemm_planes(const void* Aplanes, int lda, const float* W, int ldb, float* C, int ldc, int M, int N, int K, const float* R, int ldr, hipStream_t s, float* kpart = nullptr,
                 const LnFold* fold = nullptr, MuseWs* sk = nullptr) {
    GemmArgs g;
    g.A_hi = reinterpret_cast<const uint16_t*>(Aplanes); g.A_lo = g.A_hi + 32;
    g.B = W; g.C = C; g.R = R;
    g.M = M; g.N = N; g.K = K;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ldr = ldr;
    const bool use_sk = sk && gemm_sk_pays(M, N, K);   // (stream-K where it pays: it keeps the fused epilogue and needs no reduce launch)
    if (kpart && N <= 1024 && !fold && !use_sk) { g.ksplit = pick_ksplit(M, N, K); g.kpart = kpart; }   // the workspace holds KSPLIT_MAX slices of [rows, D] (a folded projection
    set_fold(g, fold, M);                                                                                // keeps its epilogue: the split-K reduce kernel has none of it)
    if (use_sk) set_sk(g, sk);
    launch_gemm(g, s);
}

// to_q projection with the query preparation (l2norm, q_scale, hi/lo split, head-major layout) fused into the GEMM epilogue
void gemm_planes_q(const void* Aplanes, int lda, const float* W, const float* q_scale, void* Qh, void* Ql, int B, int H, int Nq, int D, hipStream_t s, float* kpart = nullptr,
                   float* qraw = nullptr, const LnFold* fold = nullptr, MuseWs* sk = nullptr) {
    const bool use_sk = sk && gemm_sk_pays((long)B * Nq, H * 64, D);
    if (!fold && !use_sk && kpart && qraw && H * 64 <= 1024 && pick_ksplit((long)B * Nq, H * 64, D) > 1) {
        // small batch: the projection split over K into qraw, then the query preparation as its own (tiny) kernel
        gemm_planes(Aplanes, lda, W, D, qraw, H * 64, B * Nq, H * 64, D, nullptr, 0, s, kpart);
        launch_muse_q_prep_split(qraw, q_scale, Qh, Ql, B, H, Nq, 8.0f * kLog2e, s);
        return;
    }
    GemmArgs g;
    g.A_hi = reinterpret_cast<const uint16_t*>(Aplanes); g.A_lo = g.A_hi + 32;
    g.B = W;
    g.M = B * Nq; g.N = H * 64; g.K = D;
    g.lda = lda; g.ldb = D; g.ldc = H * 64;
    g.epi = EPI_MUSE_Q; g.epi_scale = q_scale; g.epi_hi = Qh; g.epi_lo = Ql; g.epi_rows = Nq; g.epi_heads = H;
    g.epi_post = 8.0f * kLog2e;   // sim = 8 q.k (muse_net:150) in the base-2 domain of the split attention kernel
    set_fold(g, fold, B * Nq);
    if (use_sk) set_sk(g, sk);
    launch_gemm(g, s);
}

void set_sk(GemmArgs& g, MuseWs* w) {
    if (!w || !w->sk_ws) return;
    g.sk_ws = w->sk_ws;
    g.sk_epoch = ++w->sk_epoch;
}

// per-batch constants: embeddings + cross-attention K/V
void muse_prepare(Ctx& c, MuseWs& w, const int64_t* cond, const float* I_inv, const float* E_inv, int B, hipStream_t s, int samples_per_layout = 1) {
    const auto& g = c.cfg;
    const std::string p = "transformer.";
    const int S = samples_per_layout < 1 ? 1 : samples_per_layout;
    BG_REQUIRE(B % S == 0, "batch %d is not a multiple of samples_per_layout %d", B, S);
    const int G = B / S;   // BEV layouts: the condition side (context embedding, cross-attention K / V of every layer) is built once per layout
    w.S = S;
    w.B = B;
    w.rows = (long)B * c.N;
    Arena& a = c.arena;
    a.reserve(muse_ws_bytes(c, B));
    a.reset();
    const int D = c.D, H = c.H;
    w.img = g.image_embed ? a.get<float>((size_t)w.rows * D) : nullptr;
    w.c_embed = g.image_embed ? a.get<float>((size_t)B * g.num_cams * D) : nullptr;
    w.context = a.get<float>((size_t)G * c.K * D);
    w.x = a.get<float>((size_t)w.rows * D);
    w.xn = a.get<float>((size_t)w.rows * D);
    w.qraw = a.get<float>((size_t)w.rows * D);
    w.att = a.get<float>((size_t)w.rows * D);
    w.kvraw = a.get<float>((size_t)std::max<long>(w.rows, (long)B * c.K) * 2 * D);
    w.Q = a.get<float>((size_t)w.rows * D);
    const size_t kvS = (size_t)B * H * c.NkS_pad * 64;
    w.Ks = a.get<float>(kvS);
    w.Vs = a.get<float>(kvS);
    w.h = a.get<float>((size_t)w.rows * 2 * c.F);
    w.g = a.get<float>((size_t)w.rows * c.Fpad);
    w.kpart = std::max(pick_ksplit(w.rows, D, D), pick_ksplit(w.rows, D, c.Fpad)) > 1 ? a.get<float>((size_t)KSPLIT_MAX * w.rows * D) : nullptr;
    w.attn_ks = pick_attn_ksplit((long)cdiv(c.N, 256) * H * B, c.NkS_pad / 32);
    w.attn_ws = w.attn_ks > 1 ? a.get<float>((size_t)attn_split_ws_floats(B, H, c.N, w.attn_ks)) : nullptr;
    w.stats_d = a.get<float>((size_t)w.rows * 2 * (D / 32));
    w.stats_f = a.get<float>((size_t)w.rows * 2 * (c.Fpad / 32));
    w.rowstat = a.get<float>((size_t)w.rows * 2);
    w.sk_ws = nullptr;
    if (w.rows <= kSkMaxRows && g.precision == BEVGEN_PRECISION_F16X3) {
        w.sk_ws = a.alloc(gemm_sk_ws_bytes());
        HIP_CHECK(hipMemsetAsync(w.sk_ws, 0, 1024 * sizeof(unsigned), s));   // the flag words: a flag equal to a launch's epoch means "published in this launch"
        w.sk_epoch = 0;
    }
    HIP_CHECK(hipMemsetAsync(w.Ks, 0, kvS * sizeof(float), s));  // rows beyond the real keys stay zero
    HIP_CHECK(hipMemsetAsync(w.Vs, 0, kvS * sizeof(float), s));

    if (g.image_embed)
        launch_camera_embed(I_inv, E_inv, c.image_plane, c.pf(p + "img_embed.weight"), c.pf(p + "cam_embed.weight"), w.img, w.c_embed, B, g.num_cams, c.T, D, s);
    const int64_t* cond_g = cond;
    const float* c_embed_g = w.c_embed;
    if (S > 1) {   // the first scene of every group -> contiguous [G, ...] inputs of the condition side
        int64_t* cg = a.get<int64_t>((size_t)G * c.K);
        HIP_CHECK(hipMemcpy2DAsync(cg, (size_t)c.K * 8, cond, (size_t)S * c.K * 8, (size_t)c.K * 8, G, hipMemcpyDeviceToDevice, s));
        cond_g = cg;
        if (w.c_embed) {
            float* eg = a.get<float>((size_t)G * g.num_cams * D);
            HIP_CHECK(hipMemcpy2DAsync(eg, (size_t)g.num_cams * D * 4, w.c_embed, (size_t)S * g.num_cams * D * 4, (size_t)g.num_cams * D * 4, G, hipMemcpyDeviceToDevice, s));
            c_embed_g = eg;
        }
    }
    launch_cond_embed(cond_g, c.pf(p + "cond_token_emb.weight"), c.pf(p + "cond_pos_emb.weight"), g.bev_embed ? c.pf(p + "bev_grid") : nullptr,
                      g.bev_embed ? c.pf(p + "bev_embed.weight") : nullptr, g.bev_embed ? c.pf(p + "bev_embed.bias") : nullptr,
                      g.bev_embed ? c.pf(p + "bev_cam_pos_emb") : nullptr, c_embed_g, w.context, G, g.num_cams, c.K, D, g.cond_vocab_size, s);

    // cross-attention keys/values: to_kv(context) (context is NOT layer-normed, muse_net:128-132), null kv prepended, k l2-normalised
    const size_t kvC = (size_t)G * H * c.NkC_pad * 64;
    w.crossK.resize(g.num_layers);
    w.crossV.resize(g.num_layers);
    for (int i = 0; i < g.num_layers; ++i) {
        w.crossK[i] = a.get<float>(kvC);
        w.crossV[i] = a.get<float>(kvC);
        HIP_CHECK(hipMemsetAsync(w.crossK[i], 0, kvC * sizeof(float), s));
        HIP_CHECK(hipMemsetAsync(w.crossV[i], 0, kvC * sizeof(float), s));
        const MuseLayer& l = c.muse[i];
        gemm(w.context, D, l.to_kv[1], D, w.kvraw, 2 * D, G * c.K, 2 * D, D, nullptr, 0, s);
        if (c.cfg.precision == BEVGEN_PRECISION_F16X3) {  // each fp32-sized buffer holds the (hi, lo) f16 planes back to back
            _Float16* kh = reinterpret_cast<_Float16*>(w.crossK[i]);
            _Float16* vh = reinterpret_cast<_Float16*>(w.crossV[i]);
            launch_muse_kv_prep_split(w.kvraw, l.null_kv[1], l.k_scale[1], kh, kh + kvC, vh, vh + kvC, G, H, c.K, c.NkC_pad, s);
        } else {
            launch_muse_kv_prep(w.kvraw, l.null_kv[1], l.k_scale[1], w.crossK[i], w.crossV[i], G, H, c.K, c.NkC_pad, s);
        }
    }
}

// one transformer pass over the current ids: leaves LayerNorm(x) (= `embed`, muse_net:202) in w.xn
void muse_blocks(Ctx& c, MuseWs& w, const int64_t* ids, hipStream_t s) {
    const auto& g = c.cfg;
    // (Round 3 built a first LayerNorm fold - producers wrote (row x gamma) planes + per-group (mean, M2), a merge kernel per LayerNorm - token-exact and SLOWER, 10.12 ->
    // 9.86 scenes/s, profiles/r03_ab_ln_fold.txt; removed in round 4.  Round 6's form (ln_fold_level above) has no merge launch - the consumer adds the group sums itself, in the
    // shadow of its first DMA - no gamma in the producer (it sits in W o gamma), and the inner LayerNorm's producer writes planes INSTEAD of its fp32 result.)
    const std::string p = "transformer.";
    const int D = c.D, H = c.H, B = w.B, N = c.N;
    const int rows = (int)w.rows;
    launch_token_embed(ids, c.pf(p + "token_emb.weight"), w.img, c.pf(p + "pos_emb.weight"), w.x, B, N, D, g.vocab_size + 1, s);
    const bool split = g.precision == BEVGEN_PRECISION_F16X3;
    const int fold = split ? ln_fold_level(c, rows) : 0;
    // fold == 2: the residual-stream projections (to_out of both attention modules, the feed-forward's down-projection) write, besides the fp32 row, the raw planes of the
    // row into w.xn and its statistics into w.stats_d: the next projection multiplies them by W o gamma.  `x_planes_ready`: w.xn / w.stats_d hold the current w.x
    bool x_planes_ready = false;
    // who merges a LayerNorm's group sums: up to two scenes every workgroup of the consuming projection runs ONE tile and merges its rows itself (no launch between producer
    // and consumer: at one scene a 9 us kernel + its ramp per LayerNorm); above that one finalize kernel per LayerNorm feeds all tiles ($BEVGEN_LN_MERGE = kernel | tile)
    const char* merge_env = getenv("BEVGEN_LN_MERGE");   // (read per forward: the tests switch it inside one process)
    const bool tile_merge = merge_env ? merge_env[0] == 't' : rows <= 3072;
    auto consumer = [&](const float* gsums, int groups, int count, const float* cs) {
        LnFold f;
        f.in_cs = cs;
        if (tile_merge) { f.in_gsums = gsums; f.in_groups = groups; f.in_count = count; }
        else { launch_ln_stats_finalize(gsums, w.rowstat, rows, groups, count, 1e-5f, s); f.in_stats = w.rowstat; }
        return f;
    };
    LnFold prod_x;   // producer role of a residual-stream projection
    prod_x.out_planes = w.xn; prod_x.out_stats = w.stats_d; prod_x.out_ld = D;
    for (int i = 0; i < g.num_layers; ++i) {
        const MuseLayer& l = c.muse[i];
        // ---- self attention
        // split-precision mode: every GEMM input is produced directly as (hi, lo) f16 planes (same bytes as the fp32 buffer they replace)
        if (split) {
            const bool f0 = fold == 2 && x_planes_ready;   // (layer 0 reads the embedding: no projection produced it - its first LayerNorm stays a kernel)
            LnFold cons_x;   // consumer role behind a LayerNorm over the D residual columns
            if (!f0) launch_layernorm_planes(w.x, D, l.norm_g[0], nullptr, w.xn, D, rows, D, 1e-5f, s);
            else cons_x = consumer(w.stats_d, D / 32, D, nullptr);
            // to_q and to_kv read the same LayerNorm planes (muse_net:126-132): ONE projection over the concatenated weight, query / key / value preparation in its
            // epilogue ($BEVGEN_QKV_MERGE=0: the two launches of rounds 2-4, for A/B runs)
            // Measured, round 5 (profiles/r05_ab_qkv_merge*.txt): one scene 161.9 -> 160.4 ms, sixteen scenes 10.31 -> 10.21 scenes/s - merged only on the low-latency path then.
            // Round 6, after the staged row-major epilogues and with the row-split rest choosing its own block shape (profiles/r06_ab_qkv_merge.txt, same box, ms per step
            // separate -> merged): three scenes 387.0 -> 373.5 (+3.6 %), four 443.7 -> 437.2, six 614.0 -> 607.1, eight 771.0 -> 768.7, twelve 1122.0 -> 1108.7 (+1.2 %),
            // sixteen 1424.0 -> 1422.2: merged at every batch size.  $BEVGEN_QKV_MERGE = 0 never, 3 only up to 3072 token rows (the round-5 rule; A/B runs)
            static const int qkv_merge = getenv("BEVGEN_QKV_MERGE") ? atoi(getenv("BEVGEN_QKV_MERGE")) : 1;
            const bool merged = l.to_qkv_self && qkv_merge != 0 && (qkv_merge != 3 || rows <= 3072);
            if (!merged) {
                cons_x.in_cs = l.fold_q_self_cs;
                gemm_planes_q(w.xn, D, f0 ? l.fold_q_self : l.to_q[0], l.q_scale[0], w.Q, reinterpret_cast<_Float16*>(w.Q) + (size_t)rows * D, B, H, N, D, s, w.kpart, w.qraw,
                              f0 ? &cons_x : nullptr, &w);
            }
            {   // to_kv with the key / value preparation in its epilogue: k planes [B, H, NkS_pad, 64], v planes transposed [B, H, 64, NkS_pad]
                const size_t kvS_ = (size_t)B * H * c.NkS_pad * 64;
                _Float16 *Kp = reinterpret_cast<_Float16*>(w.Ks), *Vp = reinterpret_cast<_Float16*>(w.Vs);
                GemmArgs gk;
                gk.A_hi = reinterpret_cast<const uint16_t*>(w.xn); gk.A_lo = gk.A_hi + 32;
                gk.B = merged ? (f0 ? l.fold_qkv : l.to_qkv_self) : (f0 ? l.fold_kv_self : l.to_kv[0]);
                gk.M = rows; gk.N = (merged ? 3 : 2) * D; gk.K = D; gk.lda = D; gk.ldb = D; gk.ldc = gk.N;
                gk.epi = merged ? EPI_MUSE_QKV : EPI_MUSE_KV;
                gk.epi_scale = l.k_scale[0]; gk.epi_hi = Kp; gk.epi_lo = Kp + kvS_; gk.epi_hi2 = Vp; gk.epi_lo2 = Vp + kvS_;
                gk.epi_aux = l.null_self; gk.epi_rows = N; gk.epi_heads = H; gk.epi_ld = c.NkS_pad;
                if (merged) {
                    gk.epi_qh = w.Q; gk.epi_ql = reinterpret_cast<_Float16*>(w.Q) + (size_t)rows * D; gk.epi_qscale = l.q_scale[0];
                    gk.epi_post = 8.0f * kLog2e;   // sim = 8 q.k (muse_net:150) in the base-2 domain of the split attention kernel
                }
                if (f0) { cons_x.in_cs = merged ? l.fold_qkv_cs : l.fold_kv_self_cs; set_fold(gk, &cons_x, rows); }
                // one scene: 144 blocks of 256 x 128 (one per CU, the staged row-major epilogue) against 288 of 128 x 128 sharing CUs in pairs on a two-stage ring
                static const int qkv_wm4 = getenv("BEVGEN_QKV_WM4") ? atoi(getenv("BEVGEN_QKV_WM4")) : 1;   // (one scene 158.9 -> 153.8 ms, profiles/r06_ab_b1_qkv_wm4.txt; 0: A/B runs)
                if (qkv_wm4 && merged) gk.force_wm = 4;
                if (gemm_sk_pays(rows, gk.N, gk.K)) set_sk(gk, &w);
                launch_gemm(gk, s);
            }
        } else {
            launch_layernorm(w.x, D, l.norm_g[0], nullptr, w.xn, D, rows, D, 1e-5f, s);
            gemm(w.xn, D, l.to_q[0], D, w.qraw, D, rows, D, D, nullptr, 0, s);
            gemm(w.xn, D, l.to_kv[0], D, w.kvraw, 2 * D, rows, 2 * D, D, nullptr, 0, s);
        }
        const size_t qN = (size_t)rows * D, kvS = (size_t)B * H * c.NkS_pad * 64, kvC = (size_t)(B / w.S) * H * c.NkC_pad * 64;
        _Float16 *Qh = reinterpret_cast<_Float16*>(w.Q), *Ksh = reinterpret_cast<_Float16*>(w.Ks), *VTsh = reinterpret_cast<_Float16*>(w.Vs);
        AttnSplitArgs sa{};
        if (split) {
            sa.Qh = Qh; sa.Ql = Qh + qN; sa.Kh = Ksh; sa.Kl = Ksh + kvS; sa.VTh = VTsh; sa.VTl = VTsh + kvS;
            sa.bias = c.bias_self; sa.bias_pk = c.bias_self_pk; sa.O = w.att; sa.B = B; sa.H = H; sa.Nq = N; sa.Nk_pad = c.NkS_pad;
            sa.ldbias = c.ldS; sa.bias_head_stride = 0; sa.scale = 8.0f * kLog2e;
            sa.o_bstride = (long)N * D; sa.o_qstride = D; sa.o_hstride = 64;
            sa.Op = reinterpret_cast<_Float16*>(w.att);
            sa.ksplit = w.attn_ks; sa.kws = w.attn_ws;
            launch_attention_split(sa, s);
            sa.ksplit = 1; sa.kws = nullptr;   // (the cross-attention's 9 key tiles stay one range)
        } else {
            launch_muse_q_prep(w.qraw, l.q_scale[0], w.Q, B, H, N, s);
            launch_muse_kv_prep(w.kvraw, l.null_kv[0], l.k_scale[0], w.Ks, w.Vs, B, H, N, c.NkS_pad, s);
        }
        AttnArgs a{};
        a.Q = w.Q; a.K = w.Ks; a.V = w.Vs; a.bias = c.bias_self; a.R = nullptr; a.O = w.att;
        a.B = B; a.H = H; a.Nq = N; a.Nk_pad = c.NkS_pad;
        a.q_bstride = (long)H * N * 64; a.q_hstride = (long)N * 64;
        a.kv_bstride = (long)H * c.NkS_pad * 64; a.kv_hstride = (long)c.NkS_pad * 64;
        a.ldbias = c.ldS; a.bias_head_stride = 0; a.scale = 8.0f;
        a.o_bstride = (long)N * D; a.o_qstride = D; a.o_hstride = 64;
        if (!split) launch_attention(a, s);
        if (split) gemm_planes(w.att, D, l.to_out[0], D, w.x, D, rows, D, D, w.x, D, s, w.kpart, fold == 2 ? &prod_x : nullptr, &w);   // x = to_out(att) + x
        else gemm(w.att, D, l.to_out[0], D, w.x, D, rows, D, D, w.x, D, s);
        // ---- cross attention
        if (split) {
            LnFold cons_x;
            if (fold != 2) launch_layernorm_planes(w.x, D, l.norm_g[1], nullptr, w.xn, D, rows, D, 1e-5f, s);
            else cons_x = consumer(w.stats_d, D / 32, D, l.fold_q_cross_cs);
            gemm_planes_q(w.xn, D, fold == 2 ? l.fold_q_cross : l.to_q[1], l.q_scale[1], w.Q, reinterpret_cast<_Float16*>(w.Q) + (size_t)rows * D, B, H, N, D, s, w.kpart, w.qraw,
                          fold == 2 ? &cons_x : nullptr, &w);
        } else {
            launch_layernorm(w.x, D, l.norm_g[1], nullptr, w.xn, D, rows, D, 1e-5f, s);
            gemm(w.xn, D, l.to_q[1], D, w.qraw, D, rows, D, D, nullptr, 0, s);
        }
        if (split) {
            _Float16 *ckh = reinterpret_cast<_Float16*>(w.crossK[i]), *cvh = reinterpret_cast<_Float16*>(w.crossV[i]);
            sa.Kh = ckh; sa.Kl = ckh + kvC; sa.VTh = cvh; sa.VTl = cvh + kvC;
            sa.bias = c.bias_cross; sa.bias_pk = c.bias_cross_pk; sa.Nk_pad = c.NkC_pad; sa.ldbias = c.ldC;
            sa.kv_group = w.S;   // the samples of a layout read the layout's cross-attention K / V
            launch_attention_split(sa, s);
        } else {
            launch_muse_q_prep(w.qraw, l.q_scale[1], w.Q, B, H, N, s);
            a.K = w.crossK[i]; a.V = w.crossV[i]; a.bias = c.bias_cross; a.Nk_pad = c.NkC_pad;
            a.kv_bstride = (long)H * c.NkC_pad * 64; a.kv_hstride = (long)c.NkC_pad * 64;
            a.ldbias = c.ldC;
            a.kv_group = w.S;
            launch_attention(a, s);
        }
        // ---- feed forward
        if (split) {
            gemm_planes(w.att, D, l.to_out[1], D, w.x, D, rows, D, D, w.x, D, s, w.kpart, fold == 2 ? &prod_x : nullptr, &w);
            if (fold != 2) launch_layernorm_planes(w.x, D, l.ff_g0, nullptr, w.xn, D, rows, D, 1e-5f, s);
            if (l.ff_w1_geglu) {
                // GEGLU in the up-projection's epilogue: h = gate * gelu(x) as [rows, Fpad] (pad columns exactly 0), then the LayerNorm half on its own - or (fold >= 1)
                // folded away: the epilogue writes the raw planes of h and its row statistics, the down-projection multiplies by W4 o gamma
                GemmArgs ge;
                ge.A_hi = reinterpret_cast<const uint16_t*>(w.xn); ge.A_lo = ge.A_hi + 32;
                ge.B = fold == 2 ? l.fold_w1 : l.ff_w1_geglu; ge.C = w.h;
                ge.M = rows; ge.N = 2 * c.Fpad; ge.K = D;
                ge.lda = D; ge.ldb = D; ge.ldc = c.Fpad;
                ge.epi = EPI_GEGLU;
                LnFold fg;
                if (fold == 2) fg = consumer(w.stats_d, D / 32, D, l.fold_w1_cs);
                if (fold >= 1) { fg.out_planes = w.g; fg.out_stats = w.stats_f; fg.out_ld = c.Fpad; }
                if (fold >= 1) set_fold(ge, &fg, rows);
                if (gemm_sk_pays(rows, ge.N, ge.K)) set_sk(ge, &w);
                launch_gemm(ge, s);
                if (fold == 0) launch_layernorm_planes(w.h, c.Fpad, l.ff_g3, nullptr, w.g, c.Fpad, rows, c.F, 1e-5f, s);
            } else {
                gemm_planes(w.xn, D, l.ff_w1, D, w.h, 2 * c.F, rows, 2 * c.F, D, nullptr, 0, s);
                launch_geglu_layernorm_planes(w.h, 2 * c.F, l.ff_g3, w.g, c.Fpad, rows, c.F, 1e-5f, s);
            }
            LnFold fd;   // the down-projection: consumer of the inner LayerNorm (fold >= 1), producer for the next layer's first LayerNorm (fold == 2, not behind the last layer)
            if (fold >= 1 && l.ff_w1_geglu) fd = consumer(w.stats_f, c.Fpad / 32, c.F, l.fold_w4_cs);
            const bool prod_next = fold == 2 && i + 1 < g.num_layers;
            if (prod_next) { fd.out_planes = w.xn; fd.out_stats = w.stats_d; fd.out_ld = D; }
            const bool cons = fd.in_stats || fd.in_gsums, any = cons || fd.out_planes;
            gemm_planes(w.g, c.Fpad, cons ? l.fold_w4 : l.ff_w4_padded, c.Fpad, w.x, D, rows, D, c.Fpad, w.x, D, s, w.kpart, any ? &fd : nullptr, &w);
            x_planes_ready = prod_next;
        } else {
            gemm(w.att, D, l.to_out[1], D, w.x, D, rows, D, D, w.x, D, s);
            launch_layernorm(w.x, D, l.ff_g0, nullptr, w.xn, D, rows, D, 1e-5f, s);
            gemm(w.xn, D, l.ff_w1, D, w.h, 2 * c.F, rows, 2 * c.F, D, nullptr, 0, s);
            launch_geglu_layernorm(w.h, 2 * c.F, l.ff_g3, w.g, c.Fpad, rows, c.F, 1e-5f, s);
            gemm(w.g, c.Fpad, l.ff_w4_padded, c.Fpad, w.x, D, rows, D, c.Fpad, w.x, D, s);
        }
    }
    launch_layernorm(w.x, D, c.pf(p + "transformer_blocks.norm.gamma"), nullptr, w.xn, D, rows, D, 1e-5f, s);
}

}  // namespace

void muse_forward(Ctx& c, const int64_t* ids, const int64_t* cond, const float* I_inv, const float* E_inv, int B, float* logits, float* embed, hipStream_t s) {
    BG_REQUIRE(c.cfg.route == BEVGEN_ROUTE_MASKGIT, "context was not created for the MaskGit route");
    BG_REQUIRE(B >= 1, "batch must be positive");
    MuseWs w;
    muse_prepare(c, w, cond, I_inv, E_inv, B, s);
    muse_blocks(c, w, ids, s);
    const int rows = (int)w.rows;
    if (embed) HIP_CHECK(hipMemcpyAsync(embed, w.xn, (size_t)rows * c.D * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (logits) gemm(w.xn, c.D, c.pf("transformer.to_logits.weight"), c.D, logits, c.V, rows, c.V, c.D, nullptr, 0, s);
}

void maskgit_generate(Ctx& c, const int64_t* cond, const float* I_inv, const float* E_inv, int B, int timesteps, const int32_t* sched, float temperature,
                      int topk_k, float critic_noise_scale, const float* gumbel_u, const float* critic_u, const int64_t* init_ids, int64_t* out, hipStream_t s,
                      unsigned long long noise_seed, int score_mode, int samples_per_layout) {
    BG_REQUIRE(c.cfg.route == BEVGEN_ROUTE_MASKGIT, "context was not created for the MaskGit route");
    BG_REQUIRE(B >= 1 && timesteps >= 1 && sched, "bad generate arguments");
    BG_REQUIRE(score_mode != 0 || c.find("token_critic.to_pred.weight"), "maskgit_generate: the context holds no token critic (token_critic.to_pred.*): use score_mode 1 / 2");
    BG_REQUIRE(score_mode >= 0 && score_mode <= 2, "score_mode %d: 0 token critic, 1 softmax confidence, 2 softmax confidence with re-masking of earlier tokens", score_mode);
    MuseWs w;
    muse_prepare(c, w, cond, I_inv, E_inv, B, s, samples_per_layout);
    const int rows = (int)w.rows;            // B*C*T token rows
    const int seqs = B * c.cfg.num_cams;     // rows of the [B*C, T] id matrix
    const int T = c.T, V = c.V, D = c.D;
    float* logits = c.arena.get<float>((size_t)rows * V);
    float* scores = c.arena.get<float>((size_t)rows);
    const int64_t mask_id = c.cfg.vocab_size;
    int64_t* ids = out;  // generation happens in the caller's output buffer
    launch_fill_i64(ids, rows, mask_id, s);
    launch_fill(scores, rows, 0.f, s);
    for (int step = 0; step < timesteps; ++step) {
        const int steps_until_x0 = timesteps - 1 - step;
        const double frac = (double)steps_until_x0 / (double)timesteps;
        launch_remask(ids, scores, init_ids, seqs, T, sched[step], mask_id, s);
        muse_blocks(c, w, ids, s);
        gemm(w.xn, D, c.pf("transformer.to_logits.weight"), D, logits, V, rows, V, D, nullptr, 0, s);
        // score_mode 1 / 2 (force_not_use_token_critic, muse_net:611-622): the scores come out of the pick itself (1 - softmax(logits)[pred]) and the critic forward
        // is not run at all - 18 transformer forwards per call instead of 35
        launch_maskgit_pick(ids, logits, V, gumbel_u ? gumbel_u + (size_t)step * rows * V : nullptr, rows, V, topk_k, (float)((double)temperature * frac), mask_id, s,
                            noise_seed, (unsigned)step, score_mode ? scores : nullptr, score_mode);
        if (score_mode == 0 && step + 1 < timesteps) {  // the critic scores only select what the NEXT iteration re-masks
            muse_blocks(c, w, ids, s);
            launch_critic_scores(w.xn, D, c.pf("token_critic.to_pred.weight"), c.pf("token_critic.to_pred.bias"),
                                 critic_u ? critic_u + (size_t)step * rows : nullptr, critic_noise_scale, (float)frac, scores, rows, D, s, noise_seed, (unsigned)step);
        }
    }
}

}  // namespace bevgen
