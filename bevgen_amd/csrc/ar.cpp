// Route A: autoregressive sparse-causal transformer with camera bias, run as prefill + KV-cache decode.
//   GPT.forward                      transformer/mingpt_sparse.py:319-391
//   Block.forward                    :240-253   (x = ln1(x); x = x + attn(x); x = x + mlp(ln2(x)) - residual from ln1(x), no out-proj)
//   CustomSparseSelfAttention        :185-212   + SparseSelfAttention.forward transformer/sparse_self_attention.py:103-177
//   Net2NetTransformer.sample        stage2/cond_transformer_multi_view.py:154-227
//
// The reference re-runs the whole L-token forward for every generated token (ar_lm:198).  Image rows are causal in decode
// order and condition rows only see condition columns (mask_generator.py:148, 202-206), so the logits of step s depend only on
// rows <= K-1+s: we compute the K condition rows once (prefill), keep K/V of every layer, and per step push exactly one new
// row through the stack (validity verified against the reference).
#include <atomic>
#include <mutex>
#include "model.h"
#include "profiler.h"

namespace bevgen {

// Decode steps that contain a launch whose workgroups wait for each other (ar_mlp_fused_kernel: every workgroup of an XCD must be resident at once) must not interleave
// with another such stream of launches on the same device: two half-resident grids could starve each other until the bounded spin gives up.  One context is one stream
// of strictly ordered launches; the decode calls of ALL Route-A contexts of the process are chained on the device by one event per device (a call's launches wait until
// the previous call's have drained) and on the host by a mutex held while a call enqueues: one uncontended lock, one stream wait and one event record per call - taken
// unconditionally (a "more than one context alive" fast path would race with a context created while another one is decoding).  Another PROCESS on the same GPU is not
// covered: there the bounded spin times out, the launch poisons itself (later launches return at once), the status word reports it and the context falls back to the
// two-launch form (Ctx::check_status).
namespace {
std::mutex g_spin_mu;
hipEvent_t g_spin_ev[64] = {};
struct SpinSerial {
    std::unique_lock<std::mutex> lk;
    hipEvent_t ev = nullptr;
    hipStream_t q = nullptr;
    SpinSerial(const Ctx& c, hipStream_t stream) {
        if (!c.mlpf_sync || c.mlpf_disabled) return;
        lk = std::unique_lock<std::mutex>(g_spin_mu);
        hipEvent_t& e = g_spin_ev[c.device >= 0 && c.device < 64 ? c.device : 0];
        if (!e) HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ev = e; q = stream;
        HIP_CHECK(hipStreamWaitEvent(q, ev, 0));   // (never recorded yet: a no-op)
    }
    void rebind(hipStream_t stream) { if (ev) { HIP_CHECK(hipStreamWaitEvent(stream, ev, 0)); q = stream; } }
    ~SpinSerial() { if (ev) (void)hipEventRecord(ev, q); }
};
}  // namespace

namespace {

void gemm(const float* A, int lda, const float* W, int ldb, const float* bias, float* C, int ldc, int M, int N, int K, int act, const float* R, int ldr, hipStream_t s) {
    GemmArgs g;
    g.A = A; g.B = W; g.C = C; g.R = R; g.bias_n = bias;
    g.M = M; g.N = N; g.K = K;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ldr = ldr; g.act = act;
    launch_gemm(g, s);
}

size_t cache_elem_bytes(const Ctx& c) { return c.cfg.kv_cache_dtype == BEVGEN_KV_F16 ? 2 : 4; }
int cache_dtype(const Ctx& c) { return c.cfg.kv_cache_dtype == BEVGEN_KV_F16 ? 1 : 0; }

struct StepWs {
    float *x, *xn, *qkv, *x2, *h, *m1, *dec_ws, *gemm_ws, *logits;
    float *x2b, *part;   // fused path: second x2 buffer (layers alternate), split-K partials of the MLP down-projection
    int64_t* tok;
    int splits;
};

bool split_supported(const Ctx& c, int B) {
    const int D = c.D;
    return skinny_fused_supported(B, 3 * D, D, true) && (c.cfg.decode_weight_dtype != BEVGEN_W_F16 || skinny_fused_f16_ok(3 * D, D, true));
}
// BEVGEN_DECODE_AUTO: the split layer for up to four sequences (layout groups) per call, else the fused one.  Same box, config 4, fp32, full decodes, ms per step
// split / fused: B = 1 0.99 / 1.28, B = 2 1.04 / 1.24, B = 3 1.14 / 1.22, B = 4 1.17 / 1.21, B = 6 1.27 / 1.21, B = 8 1.34 / 1.24, B = 16 1.47 / 1.42.  With 16-64
// workgroups per layer the fused kernel's per-head GEMV cannot pull the 12.6 MB of q/k/v weights fast enough and a workgroup's 1.2 MB K/V range streams at one CU's
// rate; the projection kernel spreads the weights over 192 workgroups and the attention-only kernel cuts each key walk into up to four ranges.
// Round 5, with both MLP projections in one launch (ar_mlp_fused_kernel, fused path only), same box, full decodes, ms per step fused / split (profiles/r05_auto_threshold.txt):
// fp16 cache + fp16 weights B = 1 0.710 / 0.829, 2 0.715 / 0.855, 4 0.727 / 0.940, 8 0.737 / 1.014 - the fused layer wins at EVERY batch size; fp32 storage
// B = 1 1.116 / 0.919, 2 1.091 / 0.945, 3 1.094 / 1.044, 4 1.094 / 1.072, 6 1.095 / 1.186: the split layer up to four sequences, as before.
int effective_decode_path(const Ctx& c, int B, int G) {
    if (c.cfg.decode_path != BEVGEN_DECODE_AUTO) return c.cfg.decode_path;
    if (c.cfg.decode_weight_dtype == BEVGEN_W_F16 && G <= 1 && c.mlpf_sync && !c.mlpf_disabled && mlp_fused_supported(B, c.D, true)) return BEVGEN_DECODE_FUSED;
    return (B / std::max(G, 1) <= 4 && split_supported(c, B)) ? BEVGEN_DECODE_SPLIT : BEVGEN_DECODE_FUSED;
}

bool fused_path(const Ctx& c, int B, int G) {
    const int path = effective_decode_path(c, B, G);
    if (path != BEVGEN_DECODE_FUSED && path != BEVGEN_DECODE_SPLIT) return false;
    const int D = c.D;
    if (path == BEVGEN_DECODE_SPLIT && !split_supported(c, B)) return false;
    if (G > 1 && c.K % 16 != 0) return false;   // the shared condition prefix must end on a 16-key chunk boundary (the per-operator path replicates it instead)
    return ar_attn_fused_supported(B, G, D, c.H) && skinny_fused_supported(B, 4 * D, D, true) && skinny_fused_supported(B, D, 4 * D, false) &&
           skinny_fused_supported(B, c.V, D, true) && ar_attn_fused_lds_bytes(G, D, (int)round_up(c.L, 4)) <= 64 * 1024 &&
           (c.cfg.decode_weight_dtype != BEVGEN_W_F16 || (skinny_fused_f16_ok(4 * D, D, true) && skinny_fused_f16_ok(D, 4 * D, false) && skinny_fused_f16_ok(c.V, D, true)));
}

size_t part_floats(const Ctx& c, int B) { return (size_t)std::max(skinny_fused_ksplit(c.D, 4 * c.D), MLP_FUSED_PLANES) * B * c.D; }

// Both MLP projections of a layer in one launch (ar_mlp_fused_kernel): the fused three-launch layer, one chain, at most 64 rows (row chunks of 16), ln2 folded, a device whose CUs hold the
// whole grid at once.  Otherwise the two skinny launches.
bool mlp_fused_step(const Ctx& c, int Bc, bool split, int chains) {
    static const int ln2_fold = getenv("BEVGEN_LN2_FOLD") ? atoi(getenv("BEVGEN_LN2_FOLD")) : 1;
    return !split && chains == 1 && ln2_fold && c.mlpf_sync && !c.mlpf_disabled && skinny_fused_ksplit(c.D, 4 * c.D) > 1 &&
           mlp_fused_supported(Bc, c.D, c.cfg.decode_weight_dtype == BEVGEN_W_F16);
}

StepWs step_ws(Ctx& c, int B) {
    // fixed layout at the start of the per-call arena (so a captured graph keeps seeing the same addresses)
    StepWs w;
    Arena& a = c.arena;
    const int D = c.D;
    w.x = a.get<float>((size_t)B * D);
    w.xn = a.get<float>((size_t)B * D);
    w.qkv = a.get<float>((size_t)B * 3 * D);
    w.x2 = a.get<float>((size_t)B * D);
    w.h = a.get<float>((size_t)B * D);
    w.m1 = a.get<float>((size_t)B * 4 * D);
    w.splits = decode_attention_splits(B, c.H, c.L);
    w.dec_ws = reinterpret_cast<float*>(a.alloc(decode_attention_ws_bytes(B, c.H, w.splits)));
    size_t gw = std::max(std::max(gemm_skinny_ws_bytes(B, 3 * D, D), gemm_skinny_ws_bytes(B, 4 * D, D)),
                         std::max(gemm_skinny_ws_bytes(B, D, 4 * D), gemm_skinny_ws_bytes(B, c.V, D)));
    w.gemm_ws = reinterpret_cast<float*>(a.alloc(gw));
    w.logits = a.get<float>((size_t)B * c.V);
    w.tok = a.get<int64_t>((size_t)B);
    w.x2b = a.get<float>((size_t)B * D);
    w.part = a.get<float>(part_floats(c, B));
    return w;
}

size_t step_ws_bytes(const Ctx& c, int B) {
    const int D = c.D;
    size_t f = (size_t)B * D * 4 + (size_t)B * 3 * D + (size_t)B * 4 * D + (size_t)B * c.V;
    const int S = decode_attention_splits(B, c.H, c.L);
    size_t gw = std::max(std::max(gemm_skinny_ws_bytes(B, 3 * D, D), gemm_skinny_ws_bytes(B, 4 * D, D)),
                         std::max(gemm_skinny_ws_bytes(B, D, 4 * D), gemm_skinny_ws_bytes(B, c.V, D)));
    return f * sizeof(float) + decode_attention_ws_bytes(B, c.H, S) + gw + B * sizeof(int64_t) + ((size_t)B * D + part_floats(c, B)) * sizeof(float) + 16 * 256;
}

void small_gemm(const float* A, int lda, const float* W, int ldb, const float* bias, float* C, int ldc, int M, int N, int K, int act, const float* R, int ldr,
                float* ws, hipStream_t s) {
    GemmArgs g;
    g.A = A; g.B = W; g.C = C; g.R = R; g.bias_n = bias;
    g.M = M; g.N = N; g.K = K;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ldr = ldr; g.act = act;
    if (M <= 64) launch_gemm_skinny_ws(g, ws, s);
    else launch_gemm(g, s);
}

}  // namespace

// ------------------------------------------------------------------------------------------------ operator seam
void sparse_self_attention_op(Ctx& c, const float* q, const float* k, const float* v, const int64_t* layout, const float* mask, const float* add, int B, int H,
                              int L, int block, float* out, hipStream_t s) {
    BG_REQUIRE(L % block == 0, "Sequence Length, %d, needs to be dividable by Block size %d!", L, block);  // ssa:54-57
    const int Lpad = (int)round_up(L, 32);
    const int nb = L / block;
    const size_t keep_b = (size_t)L * L + (size_t)H * nb * nb + (size_t)H * nb * (cdiv(L, 16) + 1) * sizeof(uint16_t) + attn_tiles_elems(H, L, Lpad) * sizeof(uint16_t) + 4 * 256;
    const size_t bias_b = (size_t)H * L * Lpad * sizeof(float);
    const size_t kv_b = Lpad != L ? (size_t)2 * B * H * Lpad * 64 * sizeof(float) : 0;
    c.arena.reserve(keep_b + bias_b + kv_b + 8 * 256);
    c.arena.reset();
    SparseVis vis;
    {
        uint8_t* allowed = c.arena.get<uint8_t>((size_t)L * L);
        uint8_t* lay = c.arena.get<uint8_t>((size_t)H * nb * nb);
        uint16_t* chunks = c.arena.get<uint16_t>((size_t)H * nb * (cdiv(L, 16) + 1));
        launch_build_allowed(mask, allowed, (long)L * L, s);
        launch_build_layout(layout, lay, chunks, H, nb, block, L, (int)cdiv(L, 16) + 1, s);
        vis.allowed = allowed; vis.ldallowed = L;
        vis.lay = lay; vis.lay_head_stride = (long)nb * nb; vis.nb = nb; vis.blk = block;
    }
    float* bias = c.arena.get<float>((size_t)H * L * Lpad);
    launch_build_masked_bias(add, vis, bias, H, L, L, Lpad, L, 0.125f, s);
    // only the key tiles in which some row of a query block sees a key are loaded and multiplied (the reference computes the nonzero layout blocks only, ssa:63-85)
    uint16_t* tiles = c.arena.get<uint16_t>(attn_tiles_elems(H, L, Lpad));
    launch_build_attn_tiles(bias, (long)L * Lpad, Lpad, H, L, Lpad, tiles, s);
    const float *kp = k, *vp = v;
    if (Lpad != L) {
        float* kpad = c.arena.get<float>((size_t)B * H * Lpad * 64);
        float* vpad = c.arena.get<float>((size_t)B * H * Lpad * 64);
        HIP_CHECK(hipMemsetAsync(kpad, 0, (size_t)B * H * Lpad * 64 * sizeof(float), s));
        HIP_CHECK(hipMemsetAsync(vpad, 0, (size_t)B * H * Lpad * 64 * sizeof(float), s));
        HIP_CHECK(hipMemcpy2DAsync(kpad, (size_t)Lpad * 256, k, (size_t)L * 256, (size_t)L * 256, (size_t)B * H, hipMemcpyDeviceToDevice, s));
        HIP_CHECK(hipMemcpy2DAsync(vpad, (size_t)Lpad * 256, v, (size_t)L * 256, (size_t)L * 256, (size_t)B * H, hipMemcpyDeviceToDevice, s));
        kp = kpad; vp = vpad;
    }
    AttnArgs a{};
    a.Q = q; a.K = kp; a.V = vp; a.bias = bias; a.R = nullptr; a.O = out;
    a.B = B; a.H = H; a.Nq = L; a.Nk_pad = Lpad;
    a.q_bstride = (long)H * L * 64; a.q_hstride = (long)L * 64;
    a.kv_bstride = (long)H * Lpad * 64; a.kv_hstride = (long)Lpad * 64;
    a.ldbias = Lpad; a.bias_head_stride = (long)L * Lpad; a.scale = 0.125f;
    a.o_bstride = (long)H * L * 64; a.o_qstride = 64; a.o_hstride = (long)L * 64;  // [B,H,L,64] like the reference op
    a.tiles = tiles; a.tiles_head_stride = (long)cdiv(L, 128) * (Lpad / 32 + 1); a.tiles_ld = Lpad / 32 + 1;
    launch_attention(a, s);
}

// ------------------------------------------------------------------------------------------------ prefill
// samples_per_layout = S > 1 (BASELINE config 5): consecutive groups of S sequences share their condition (BEV ids and cameras), so the condition
// prefix is pushed through the stack once per LAYOUT (B / S sequences) and its K/V rows are then replicated into the S cache slots of the group.
void ar_prefill(Ctx& c, const int64_t* cond, const float* I_inv, const float* E_inv, int B, hipStream_t s, int samples_per_layout) {
    const auto& g = c.cfg;
    BG_REQUIRE(g.route == BEVGEN_ROUTE_AR, "context was not created for the autoregressive route");
    BG_REQUIRE(B >= 1, "batch must be positive");
    const int S = samples_per_layout < 1 ? 1 : samples_per_layout;
    BG_REQUIRE(B % S == 0, "batch %d is not a multiple of samples_per_layout %d", B, S);
    const int G = B / S;   // layouts = sequences actually prefilled
    const int D = c.D, H = c.H, K = c.K, L = c.L;
    auto& st = c.ars;
    // decode_path = auto: the split layer's QKV operand image is packed by the first batch that will run it (S sequences per workgroup only on the fused paths)
    if (effective_decode_path(c, B, fused_path(c, B, S) ? S : 1) == BEVGEN_DECODE_SPLIT) ctx_pack_split_qkv(c, s);
    // persistent per-batch state
    const size_t cache_b = (size_t)g.num_layers * B * H * L * 64 * cache_elem_bytes(c);
    const size_t img_b = g.image_embed ? (size_t)B * g.num_cams * c.T * D * sizeof(float) : 0;
    const size_t need = 2 * cache_b + img_b + (size_t)B * g.num_cams * D * sizeof(float) + (size_t)B * D * sizeof(float) + 16 * 256;
    c.persist.reserve(need);
    c.persist.reset();
    st.B = B;
    st.step = 0;
    st.kcache = c.persist.alloc(cache_b);
    st.vcache = c.persist.alloc(cache_b);
    st.img_embed = g.image_embed ? reinterpret_cast<float*>(c.persist.alloc(img_b)) : nullptr;
    st.c_embed = c.persist.get<float>((size_t)B * g.num_cams * D);
    st.hidden = c.persist.get<float>((size_t)B * D);
    st.d_step = c.persist.get<int>(1);
    HIP_CHECK(hipMemsetAsync(st.kcache, 0, cache_b, s));
    HIP_CHECK(hipMemsetAsync(st.vcache, 0, cache_b, s));
    HIP_CHECK(hipMemsetAsync(st.d_step, 0, sizeof(int), s));

    // workspace: the decode-step buffers first (stable addresses), then the prefill activations
    const size_t rows = (size_t)G * K;
    const size_t tmp_kv = cache_dtype(c) ? (size_t)2 * G * H * c.Kpad * 64 : 0;   // fp16 cache: the prefill attention reads exact fp32 K/V from a scratch pair
    const size_t layer_bias = c.prefill_bias ? 0 : (size_t)c.keep_heads * K * c.Kpad;   // per-layer layouts: this layer's masked bias is built on the fly
    const size_t pre_b = (rows * D * 4 + rows * 3 * D + rows * 4 * D + (size_t)G * H * K * 64 + tmp_kv + layer_bias) * sizeof(float) + (size_t)G * (K * 8 + (size_t)(g.num_cams + 1) * D * 4) +
                         attn_tiles_elems(c.keep_heads, K, c.Kpad) * sizeof(uint16_t) + 34 * 256;
    c.arena.reserve(step_ws_bytes(c, B) + pre_b);
    c.arena.reset();
    (void)step_ws(c, B);
    float* x = c.arena.get<float>(rows * D);
    float* xn = c.arena.get<float>(rows * D);
    float* x2 = c.arena.get<float>(rows * D);
    float* h = c.arena.get<float>(rows * D);
    float* qkv = c.arena.get<float>(rows * 3 * D);
    float* m1 = c.arena.get<float>(rows * 4 * D);
    float* Q = c.arena.get<float>((size_t)G * H * K * 64);
    float* lbias = c.prefill_bias ? nullptr : c.arena.get<float>(layer_bias);
    uint16_t* ltiles = c.arena.get<uint16_t>(attn_tiles_elems(c.keep_heads, K, c.Kpad));   // key tiles of the condition rows that hold a present block (rebuilt per layer when layouts differ)
    if (c.prefill_bias) launch_build_attn_tiles(c.prefill_bias, (long)K * c.Kpad, c.Kpad, c.keep_heads, K, c.Kpad, ltiles, s);

    if (g.image_embed)   // per-sequence state of the decode steps: for all B sequences
        launch_camera_embed(I_inv, E_inv, c.image_plane, c.pf("img_embed.weight"), c.pf("cam_embed.weight"), st.img_embed, st.c_embed, B, g.num_cams, c.T, D, s);
    const int64_t* cond_g = cond;
    const float* c_embed_g = st.c_embed;
    if (S > 1) {   // first sequence of every group -> contiguous [G, ...] inputs of the prefill
        int64_t* cg = c.arena.get<int64_t>((size_t)G * K);
        float* eg = c.arena.get<float>((size_t)G * g.num_cams * D);
        HIP_CHECK(hipMemcpy2DAsync(cg, (size_t)K * 8, cond, (size_t)S * K * 8, (size_t)K * 8, G, hipMemcpyDeviceToDevice, s));
        HIP_CHECK(hipMemcpy2DAsync(eg, (size_t)g.num_cams * D * 4, st.c_embed, (size_t)S * g.num_cams * D * 4, (size_t)g.num_cams * D * 4, G, hipMemcpyDeviceToDevice, s));
        cond_g = cg; c_embed_g = eg;
    }
    launch_cond_embed(cond_g, c.pf("cond_tok_emb.weight"), c.pf("cond_pos_emb"), g.bev_embed ? c.pf("bev_grid") : nullptr, g.bev_embed ? c.pf("bev_embed.weight") : nullptr,
                      g.bev_embed ? c.pf("bev_embed.bias") : nullptr, g.bev_embed ? c.pf("bev_cam_pos_emb") : nullptr, c_embed_g, x, G, g.num_cams, K, D,
                      g.cond_vocab_size, s);
    const int kvd = cache_dtype(c);
    float *tk = nullptr, *tv = nullptr;
    if (kvd) {
        tk = c.arena.get<float>((size_t)G * H * c.Kpad * 64);
        tv = c.arena.get<float>((size_t)G * H * c.Kpad * 64);
        HIP_CHECK(hipMemsetAsync(tk, 0, (size_t)G * H * c.Kpad * 64 * sizeof(float), s));   // pad rows K..Kpad stay zero
        HIP_CHECK(hipMemsetAsync(tv, 0, (size_t)G * H * c.Kpad * 64 * sizeof(float), s));
    }
    const size_t layer_bytes_p = (size_t)B * H * L * 64 * cache_elem_bytes(c);
    for (int i = 0; i < g.num_layers; ++i) {
        const ArLayer& l = c.ar[i];
        char* kcb = reinterpret_cast<char*>(st.kcache) + i * layer_bytes_p;
        char* vcb = reinterpret_cast<char*>(st.vcache) + i * layer_bytes_p;
        float* kc = kvd ? tk : reinterpret_cast<float*>(kcb);   // what the prefill attention reads
        float* vc = kvd ? tv : reinterpret_cast<float*>(vcb);
        const int Lk = kvd ? c.Kpad : L;                        // row capacity per (sequence, head) of that buffer
        launch_layernorm(x, D, l.ln1_w, l.ln1_b, xn, D, (int)rows, D, 1e-5f, s);
        gemm(xn, D, l.wqkv, D, l.bqkv, qkv, 3 * D, (int)rows, 3 * D, D, ACT_NONE, nullptr, 0, s);
        launch_ar_qkv_scatter(qkv, Q, kc, vc, 0, G, H, K, 0, Lk, s);   // cache slots [0, G) for now
        if (kvd) launch_ar_qkv_scatter(qkv, nullptr, kcb, vcb, kvd, G, H, K, 0, L, s);   // the fp16 image the decode steps read
        AttnArgs a{};
        if (lbias) {
            launch_build_masked_bias(c.attn_bias, c.vis_of_layer(i), lbias, c.keep_heads, K, K, c.Kpad, c.L, 0.125f, s);
            launch_build_attn_tiles(lbias, (long)K * c.Kpad, c.Kpad, c.keep_heads, K, c.Kpad, ltiles, s);
        }
        a.Q = Q; a.K = kc; a.V = vc; a.bias = lbias ? lbias : c.prefill_bias; a.R = xn; a.O = x2;
        a.B = G; a.H = H; a.Nq = K; a.Nk_pad = c.Kpad;
        a.q_bstride = (long)H * K * 64; a.q_hstride = (long)K * 64;
        a.kv_bstride = (long)H * Lk * 64; a.kv_hstride = (long)Lk * 64;
        a.ldbias = c.Kpad; a.bias_head_stride = c.keep_heads > 1 ? (long)K * c.Kpad : 0; a.scale = 0.125f;
        a.tiles = ltiles; a.tiles_head_stride = c.keep_heads > 1 ? (long)cdiv(K, 128) * (c.Kpad / 32 + 1) : 0; a.tiles_ld = c.Kpad / 32 + 1;
        a.o_bstride = (long)K * D; a.o_qstride = D; a.o_hstride = 64;
        BG_REQUIRE(c.Kpad <= L, "condition length padded to %d exceeds the cache length %d", c.Kpad, L);
        launch_attention(a, s);  // x2 = ln1(x) + attn
        launch_layernorm(x2, D, l.ln2_w, l.ln2_b, h, D, (int)rows, D, 1e-5f, s);
        gemm(h, D, l.mlp0_w, D, l.mlp0_b, m1, 4 * D, (int)rows, 4 * D, D, ACT_GELU, nullptr, 0, s);
        gemm(m1, 4 * D, l.mlp2_w, 4 * D, l.mlp2_b, x, D, (int)rows, D, 4 * D, ACT_NONE, x2, D, s);
    }
    st.G = 1;
    if (S == 1) {
        launch_gather_rows(x, st.hidden, B, K - 1, K, D, s);
        return;
    }
    if (fused_path(c, B, S)) {
        // the fused decode kernel reads the K prefix rows of a group from the group's FIRST cache slot: move slot g -> slot g*S, highest group first
        // (a destination never holds a source that is still needed); the other slots of the group keep only their private rows >= K
        st.G = S;
        for (int gi = G - 1; gi >= 1; --gi)
            launch_replicate_prefix(st.kcache, st.vcache, g.num_layers, B, H, L, K, gi, gi * S, 1, (int)cache_elem_bytes(c), s);
    } else {
        // per-operator path: fan the shared prefix out into every slot of the group
        for (int gi = G - 1; gi >= 0; --gi)
            launch_replicate_prefix(st.kcache, st.vcache, g.num_layers, B, H, L, K, gi, gi * S, S, (int)cache_elem_bytes(c), s);
    }
    float* hid = c.arena.get<float>((size_t)G * D);
    launch_gather_rows(x, hid, G, K - 1, K, D, s);
    for (int j = 0; j < S; ++j)
        HIP_CHECK(hipMemcpy2DAsync(st.hidden + (size_t)j * D, (size_t)S * D * 4, hid, (size_t)D * 4, (size_t)D * 4, G, hipMemcpyDeviceToDevice, s));
}

// ln_f + head on the newest rows (mingpt_sparse.py:386-387)
static void head_logits(Ctx& c, StepWs& w, int B, float* logits, hipStream_t s) {
    auto& st = c.ars;
    if (fused_path(c, B, st.G)) {
        SkinnyFusedArgs g;
        g.A = st.hidden; g.lda = c.D;
        g.ln_w = c.pf("ln_f.weight"); g.ln_b = c.pf("ln_f.bias"); g.eps = 1e-5f;
        g.Wp = c.head_wp; g.w_f16 = c.cfg.decode_weight_dtype == BEVGEN_W_F16;
        g.C = logits; g.ldc = c.V;
        g.M = B; g.N = c.V; g.K = c.D; g.ksplit = 1;
        launch_skinny_fused(g, s);
        return;
    }
    launch_layernorm(st.hidden, c.D, c.pf("ln_f.weight"), c.pf("ln_f.bias"), w.xn, c.D, B, c.D, 1e-5f, s);
    small_gemm(w.xn, c.D, c.pf("head.weight"), c.D, nullptr, logits, c.V, B, c.V, c.D, ACT_NONE, nullptr, 0, w.gemm_ws, s);
}

void ar_logits(Ctx& c, float* logits, hipStream_t s) {
    auto& st = c.ars;
    BG_REQUIRE(st.B > 0, "bevgen_ar_prefill must be called first");
    c.arena.reset();
    StepWs w = step_ws(c, st.B);
    head_logits(c, w, st.B, logits, s);
}

// Number of independent sequence groups ("chains") the fused step is cut into.  A decode step is a chain of ~75 short, strictly dependent kernels; each one pays its ramp,
// one cold weight burst and its tail alone (the weight-streaming projections reach ~2 TB/s while they run, the chip idles in between).  Sequences are independent, so
// the batch is cut into chains enqueued on separate streams: the hardware then always has another chain's kernel to run - a projection's 16.8 MB weight burst under the
// other chain's K/V stream.  The price: every chain streams the layer's projection weights (50 MB / layer with fp32 storage) itself; the second read of a matrix
// follows the first within a layer and is served by the memory-side cache.  (A persistent single-launch layer was measured instead and rejected: a grid-wide
// exchange between phases costs 7.9 us on this 8-XCD part, profiles/r03_xchg_probe.txt - more than a kernel boundary.)
static int decode_chains(const Ctx& c, int B, int G) {
    int n = c.cfg.decode_chains;
    if (n <= 0) n = 1;   // measured (profiles/r03_chain_probe.txt, B = 16): 1 chain 1.41 ms/step, 2 chains 1.60, 4 chains 1.96 - the option stays for experiments
    n = std::min(n, 4);
    while (n > 1 && ((B / G) % n != 0 || (B / G) / n < 1)) --n;
    if (c.trace) n = 1;   // the phase-timestamp buffers are indexed by workgroup of ONE launch per kind
    return n;
}

// One chain = the sequences [r0, r0 + Bc) through all layers.  Fused form of the step (decode_fused.hip): per layer {ln1 + qkv + attention, ln2 + MLP up + GELU,
// MLP down split over K}; the down-projection's partial sums, bias and residual are folded into the next consumer's row fetch (RowSrc), the last layer's into
// the hidden-state write.
static void decode_chain_launch(Ctx& c, StepWs& w, const int64_t* tok, int r0, int Bc, int* counter, hipStream_t s) {
    const auto& g = c.cfg;
    auto& st = c.ars;
    const int D = c.D, H = c.H, B = st.B, L = c.L;
    const size_t eb = cache_elem_bytes(c);
    float* x = w.x + (size_t)r0 * D;
    if (!(st.pick_embeds && tok == w.tok && r0 == 0 && Bc == B))   // (ar_sample's pick launch already wrote the new rows' embeddings)
    launch_ar_step_embed(tok + r0, c.pf("x_tok_emb.weight"), st.img_embed ? st.img_embed + (size_t)r0 * g.num_cams * c.T * D : nullptr, c.pf("x_pos_emb"), c.fwd_idx,
                         st.d_step, x, Bc, g.num_cams, c.T, D, g.vocab_size + 1, s);
    const size_t layer_bytes = (size_t)B * H * L * 64 * eb;
    const size_t chain_off = (size_t)r0 * H * L * 64 * eb;    // this chain's first cache slot inside a layer
    const int ks = skinny_fused_ksplit(D, 4 * D);
    const int wf16 = g.decode_weight_dtype == BEVGEN_W_F16;
    float* part = w.part + (size_t)ks * r0 * D;               // [ks][Bc][D] per chain, chains back to back
    float* m1 = w.m1 + (size_t)r0 * 4 * D;
    const bool split = effective_decode_path(c, B, st.G) == BEVGEN_DECODE_SPLIT;
    const bool mlpf = mlp_fused_step(c, Bc, split, Bc == B ? 1 : 2);
    float* qkv = w.qkv + (size_t)r0 * 3 * D;
    float* xn = w.xn + (size_t)r0 * D;
    RowSrc src;
    src.base = x; src.ld = D;
    for (int i = 0; i < g.num_layers; ++i) {
        const ArLayer& l = c.ar[i];
        float* x2 = ((i & 1) ? w.x2b : w.x2) + (size_t)r0 * D;
        ArAttnFusedArgs a;
        if (split) {   // LayerNorm + QKV projection of the chain's rows in one MFMA kernel (the weight is read once), ln1(x) kept for the residual
            SkinnyFusedArgs pq;
            pq.a_src = &src;
            pq.ln_w = l.ln1_w; pq.ln_b = l.ln1_b; pq.eps = 1e-5f;
            pq.Wp = l.wqkv_wp; pq.w_f16 = wf16; pq.bias = l.bqkv;
            pq.C = qkv; pq.ldc = 3 * D;
            pq.xn_out = xn; pq.ldxn = D;
            pq.M = Bc; pq.N = 3 * D; pq.K = D; pq.ksplit = 1;
            launch_skinny_fused(pq, s);
            a.qkv = qkv; a.xn = xn;
            // one or two sequences: 16-32 workgroups cannot pull a 1.2 MB K/V range each at the chip's rate - split the key walk over more of them (the partial
            // states live in the per-operator path's workspace, sized for at least this many splits)
            if (st.G == 1 && Bc * H < 128) {
                const int ks = std::min(std::min(4, 128 / (Bc * H)), w.splits);
                if (ks > 1) { a.ksplit = ks; a.kws = w.dec_ws + (size_t)r0 * H * w.splits * 66; }   // (per chain: its sequences' slots)
            }
        } else {
            a.x = src;
            a.ln_w = l.ln1_w; a.ln_b = l.ln1_b; a.eps = 1e-5f;
            a.wqkv = l.wqkv; a.bqkv = l.bqkv; a.wqkv_h = l.wqkv_h;
            a.ln_cs = l.ln1_cs; a.ln_ds = l.ln1_ds;
        }
        a.kcache = reinterpret_cast<char*>(st.kcache) + i * layer_bytes + chain_off;
        a.vcache = reinterpret_cast<char*>(st.vcache) + i * layer_bytes + chain_off;
        a.kv_dtype = cache_dtype(c);
        a.bias = c.attn_bias; a.ldbias = L;
        a.vis = c.vis_of_layer(i);
        a.out = x2; a.ldo = D;
        a.B = Bc; a.G = st.G; a.H = H; a.D = D; a.Lmax = L;
        a.n = c.K + 1; a.d_n = st.d_step; a.n_hint = st.step;
        a.prefix = c.K; a.scale = 0.125f;
        a.trace = c.trace;
        launch_ar_attn_fused(a, s);
        if (mlpf) {   // ln2 + MLP up + GELU + MLP down in one launch: 8 partial planes (one per XCD) for the next consumer's row fetch
            MlpFusedArgs mf;
            mf.A = x2; mf.lda = D;
            mf.ln_w = l.ln2_w; mf.ln_cs = l.mlp0_cs; mf.ln_ds = l.mlp0_ds; mf.eps = 1e-5f;
            mf.Wup = l.mlp0_wp; mf.Wdn = l.mlp2_wp; mf.w_f16 = wf16;
            mf.hidden = m1; mf.C = part;
            mf.sync = c.mlpf_sync; mf.err = c.status_dev;
            mf.M = Bc; mf.D = D;
            mf.trace = c.trace ? c.trace + 4096 * 8 : nullptr;
            launch_ar_mlp_fused(mf, s);
            src = RowSrc{};
            src.base = x2; src.ld = D; src.partial = part; src.ns = MLP_FUSED_PLANES; src.pstride = (long)Bc * D; src.pld = D; src.bias = l.mlp2_b;
            continue;
        }
        SkinnyFusedArgs up;
        up.A = x2; up.lda = D;
        up.ln_w = l.ln2_w; up.ln_b = l.ln2_b; up.eps = 1e-5f;
        up.Wp = l.mlp0_wp; up.w_f16 = wf16; up.bias = l.mlp0_b;
        // ln2 folded into the product (skinny_fused_kernel<.., FD>).  Same-box A/B, ms per decode step, folded vs direct: fp16 cache + fp16 weights 0.989-0.994 vs
        // 0.992-0.998, fp16 cache + fp32 weights 1.137-1.151 vs 1.154-1.163, fp32 both 1.447-1.454 vs 1.466-1.482 (profiles/r04_ab_ln2_fold.txt).  $BEVGEN_LN2_FOLD=0: direct form
        static const int ln2_fold = getenv("BEVGEN_LN2_FOLD") ? atoi(getenv("BEVGEN_LN2_FOLD")) : 1;
        if (ln2_fold) { up.ln_cs = l.mlp0_cs; up.ln_ds = l.mlp0_ds; }
        up.C = m1; up.ldc = 4 * D;
        up.M = Bc; up.N = 4 * D; up.K = D; up.ksplit = 1; up.act = ACT_GELU;
        up.trace = c.trace ? c.trace + 4096 * 8 : nullptr;
        launch_skinny_fused(up, s);
        SkinnyFusedArgs dn;
        dn.A = m1; dn.lda = 4 * D;
        dn.Wp = l.mlp2_wp; dn.w_f16 = wf16;
        dn.M = Bc; dn.N = D; dn.K = 4 * D; dn.ksplit = ks;
        dn.trace = c.trace ? c.trace + 2 * 4096 * 8 : nullptr;
        if (ks > 1) {
            dn.C = part;
            src = RowSrc{};
            src.base = x2; src.ld = D; src.partial = part; src.ns = ks; src.pstride = (long)Bc * D; src.pld = D; src.bias = l.mlp2_b;
        } else {   // narrow models: the whole K fits one workgroup; bias here, residual through the row source
            float* hrow = w.h + (size_t)r0 * D;
            dn.C = hrow; dn.ldc = D; dn.bias = l.mlp2_b;
            src = RowSrc{};
            src.base = x2; src.ld = D; src.partial = hrow; src.ns = 1; src.pstride = 0; src.pld = D;
        }
        launch_skinny_fused(dn, s);
    }
    launch_rowsrc_materialize(src, st.hidden + (size_t)r0 * D, Bc, D, counter, s);   // hidden state of the chain's new rows (+ the step counter when this is the only chain)
}

static void decode_step_launch_fused(Ctx& c, StepWs& w, const int64_t* tok, hipStream_t s) {
    auto& st = c.ars;
    const int B = st.B;
    const int nch = decode_chains(c, B, st.G);
    if (nch == 1) {
        decode_chain_launch(c, w, tok, 0, B, st.d_step, s);
        return;
    }
    // fork: chains 1.. on side streams behind everything already enqueued on s; join before the step counter moves (every kernel of the step reads it)
    while ((int)c.chain_streams.size() < nch - 1) {
        hipStream_t q;
        hipEvent_t e;
        HIP_CHECK(hipStreamCreateWithFlags(&q, hipStreamNonBlocking));
        HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        c.chain_streams.push_back(q);
        c.chain_join.push_back(e);
    }
    if (!c.chain_fork) HIP_CHECK(hipEventCreateWithFlags(&c.chain_fork, hipEventDisableTiming));
    HIP_CHECK(hipEventRecord(c.chain_fork, s));
    const int Bc = B / nch;
    for (int k = 1; k < nch; ++k) {
        hipStream_t q = c.chain_streams[k - 1];
        HIP_CHECK(hipStreamWaitEvent(q, c.chain_fork, 0));
        decode_chain_launch(c, w, tok, k * Bc, Bc, nullptr, q);
        HIP_CHECK(hipEventRecord(c.chain_join[k - 1], q));
    }
    decode_chain_launch(c, w, tok, 0, Bc, nullptr, s);
    for (int k = 1; k < nch; ++k) HIP_CHECK(hipStreamWaitEvent(s, c.chain_join[k - 1], 0));
    launch_increment(st.d_step, s);
}

// one new row through all layers; position / bias row / cache slot are derived from the device-side step counter
static void decode_step_launch(Ctx& c, StepWs& w, const int64_t* tok, hipStream_t s) {
    const auto& g = c.cfg;
    auto& st = c.ars;
    if (fused_path(c, st.B, st.G)) {
        decode_step_launch_fused(c, w, tok, s);
        return;
    }
    const int D = c.D, H = c.H, B = st.B, L = c.L;
    launch_ar_step_embed(tok, c.pf("x_tok_emb.weight"), st.img_embed, c.pf("x_pos_emb"), c.fwd_idx, st.d_step, w.x, B, g.num_cams, c.T, D, g.vocab_size + 1, s);
    const size_t layer_bytes = (size_t)B * H * L * 64 * cache_elem_bytes(c);
    const float* xin = w.x;
    for (int i = 0; i < g.num_layers; ++i) {
        const ArLayer& l = c.ar[i];
        char* kc = reinterpret_cast<char*>(st.kcache) + i * layer_bytes;
        char* vc = reinterpret_cast<char*>(st.vcache) + i * layer_bytes;
        launch_layernorm(xin, D, l.ln1_w, l.ln1_b, w.xn, D, B, D, 1e-5f, s);
        small_gemm(w.xn, D, l.wqkv, D, l.bqkv, w.qkv, 3 * D, B, 3 * D, D, ACT_NONE, nullptr, 0, w.gemm_ws, s);
        DecodeAttnArgs a;
        a.q = w.qkv; a.ldq = 3 * D;
        a.append_k = w.qkv + D; a.append_v = w.qkv + 2 * D;  // the new row (sequence position K + step) is appended inside the attention kernel
        a.kcache = kc; a.vcache = vc;
        a.bias = c.attn_bias; a.ldbias = L;
        a.vis = c.vis_of_layer(i);
        a.R = w.xn; a.ldr = D; a.O = w.x2; a.ldo = D;
        a.B = B; a.H = H; a.n = c.K + 1; a.d_n = st.d_step; a.n_hint = st.step; a.Lmax = L;
        a.scale = 0.125f; a.kv_dtype = cache_dtype(c);
        launch_decode_attention_ws(a, w.dec_ws, w.splits, s);
        launch_layernorm(w.x2, D, l.ln2_w, l.ln2_b, w.h, D, B, D, 1e-5f, s);
        small_gemm(w.h, D, l.mlp0_w, D, l.mlp0_b, w.m1, 4 * D, B, 4 * D, D, ACT_GELU, nullptr, 0, w.gemm_ws, s);
        small_gemm(w.m1, 4 * D, l.mlp2_w, 4 * D, l.mlp2_b, w.x, D, B, D, 4 * D, ACT_NONE, w.x2, D, w.gemm_ws, s);
        xin = w.x;
    }
    HIP_CHECK(hipMemcpyAsync(st.hidden, w.x, (size_t)B * D * sizeof(float), hipMemcpyDeviceToDevice, s));
    launch_increment(st.d_step, s);
}

void ar_decode_step(Ctx& c, const int64_t* tok, hipStream_t s) {
    auto& st = c.ars;
    BG_REQUIRE(st.B > 0, "bevgen_ar_prefill must be called first");
    BG_REQUIRE(st.step < c.N, "all %d image tokens have already been decoded", c.N);
    c.arena.reset();
    StepWs w = step_ws(c, st.B);
    st.pick_embeds = false;   // the caller's tokens: embed them here
    SpinSerial serial(c, s);
    decode_step_launch(c, w, tok, s);
    st.step += 1;
}

void ar_sample(Ctx& c, const int64_t* cond, const float* I_inv, const float* E_inv, int B, int steps, int top_k, float temperature, int greedy,
               const float* noise_u, int samples_per_layout, const int64_t* forced, int64_t* out, float* step_logits, hipStream_t s) {
    BG_REQUIRE(steps >= 1 && steps <= c.N, "steps=%d out of range [1,%d]", steps, c.N);
    BG_REQUIRE(greedy || noise_u, "stochastic sampling needs explicit uniform noise d_noise_u [steps, B]");
    ar_prefill(c, cond, I_inv, E_inv, B, s, samples_per_layout);
    auto& st = c.ars;
    c.step_events_used = 0;   // bevgen_ar_step_times describes THIS call: a call that takes the eager path below reports no steps
    c.arena.reset();
    StepWs w = step_ws(c, B);
    launch_fill_i64(out, (long)B * c.N, c.cfg.vocab_size, s);  // x = vocab_size everywhere (ar_lm:157)

    // one decode iteration: logits of the newest row -> token -> (optionally) push the token through the stack.
    // Every position-dependent quantity (decode-order index, cache slot, bias row, context length, noise row) is read from the device-side
    // step counter, so the launch sequence is identical for every step and can be captured once and replayed as a hipGraph.
    // (the token's store into `out` and - fused path, one chain - the new row's embedding ride in the pick launch: two launches per step less)
    ArPickTail tail;
    tail.out_all = out; tail.fwd_idx = c.fwd_idx; tail.N = c.N;
    st.pick_embeds = fused_path(c, B, st.G) && decode_chains(c, B, st.G) == 1 && !(getenv("BEVGEN_PICK_TAIL") && atoi(getenv("BEVGEN_PICK_TAIL")) == 0);
    if (st.pick_embeds) {
        tail.tok_emb = c.pf("x_tok_emb.weight"); tail.img_embed = st.img_embed; tail.pos_emb = c.pf("x_pos_emb"); tail.x = w.x;
        tail.C = c.cfg.num_cams; tail.T = c.T; tail.D = c.D; tail.vocab_rows = c.cfg.vocab_size + 1;
    }
    auto head_and_pick = [&](float* lg, hipStream_t q) {
        head_logits(c, w, B, lg, q);
        launch_ar_pick(lg, c.V, greedy ? nullptr : noise_u, st.d_step, forced, w.tok, B, c.V, top_k, temperature, q, &tail);
    };

    const bool use_graph = steps > 2 && !step_logits && !prof_enabled() && !c.disable_graphs;
    SpinSerial serial(c, s);
    if (!use_graph) {
        for (int step = 0; step < steps; ++step) {
            head_and_pick(step_logits ? step_logits + (size_t)step * B * c.V : w.logits, s);
            if (step + 1 < steps) {
                decode_step_launch(c, w, w.tok, s);
                st.step += 1;
            }
        }
        return;
    }

    // graph path: capture {head, pick, store, one full decode step} on a library-owned stream (the caller's stream may be the legacy
    // default stream, which cannot be captured), replay it steps-1 times, finish with one eager head+pick.
    if (!c.graph_stream) {
        HIP_CHECK(hipStreamCreateWithFlags(&c.graph_stream, hipStreamNonBlocking));
        HIP_CHECK(hipEventCreateWithFlags(&c.graph_ev_in, hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&c.graph_ev_out, hipEventDisableTiming));
    }
    hipStream_t q = c.graph_stream;
    HIP_CHECK(hipEventRecord(c.graph_ev_in, s));
    HIP_CHECK(hipStreamWaitEvent(q, c.graph_ev_in, 0));
    serial.q = s;   // (the call's last launches rejoin the caller's stream below: the chain event is recorded there)
    Ctx::GraphKey key;
    key.B = B; key.G = st.G; key.chains = fused_path(c, B, st.G) ? decode_chains(c, B, st.G) : 1; key.top_k = top_k; key.greedy = greedy; key.kv = cache_dtype(c); key.temperature = temperature;
    key.noise = greedy ? nullptr : noise_u; key.forced = forced; key.out = out; key.arena = c.arena.base; key.persist = c.persist.base; key.trace = c.trace;
    if (!c.graph_exec || !(c.graph_key == key)) {
        if (c.graph_exec) c.retire_graph(c.graph_exec, c.graph);
        c.graph_exec = nullptr; c.graph = nullptr;
        hipGraph_t graph = nullptr;
        HIP_CHECK(hipStreamBeginCapture(q, hipStreamCaptureModeThreadLocal));
        try {
            head_and_pick(w.logits, q);
            decode_step_launch(c, w, w.tok, q);
        } catch (...) {
            (void)hipStreamEndCapture(q, &graph);
            if (graph) (void)hipGraphDestroy(graph);
            throw;
        }
        HIP_CHECK(hipStreamEndCapture(q, &graph));
        HIP_CHECK(hipGraphInstantiate(&c.graph_exec, graph, nullptr, nullptr, 0));
        c.graph = graph;
        c.graph_key = key;
    }
    auto stamp = [&]() {
        if (!c.time_steps) return;
        if (c.step_events_used == (int)c.step_events.size()) {
            hipEvent_t e;
            HIP_CHECK(hipEventCreate(&e));
            c.step_events.push_back(e);
        }
        HIP_CHECK(hipEventRecord(c.step_events[c.step_events_used++], q));
    };
    stamp();
    for (int step = 0; step + 1 < steps; ++step) {
        HIP_CHECK(hipGraphLaunch(c.graph_exec, q));
        stamp();
    }
    st.step += steps - 1;
    head_and_pick(w.logits, q);
    HIP_CHECK(hipEventRecord(c.graph_ev_out, q));
    HIP_CHECK(hipStreamWaitEvent(s, c.graph_ev_out, 0));
}

}  // namespace bevgen

namespace bevgen {
// durations of the graph replays of the most recent bevgen_ar_sample (one decode step each), in milliseconds; synchronises the library stream
int ar_step_times(Ctx& c, float* out_ms, int cap) {
    if (c.step_events_used < 2) return 0;
    HIP_CHECK(hipEventSynchronize(c.step_events[c.step_events_used - 1]));
    const int n = std::min(cap, c.step_events_used - 1);
    for (int i = 0; i < n; ++i) HIP_CHECK(hipEventElapsedTime(&out_ms[i], c.step_events[i], c.step_events[i + 1]));
    return n;
}
}  // namespace bevgen
