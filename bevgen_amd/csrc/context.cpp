// Context lifecycle: tensor registry (reference state_dict names), derived device data built in bevgen_finalize.
#include <cstring>

#include "model.h"

namespace bevgen {

// ------------------------------------------------------------------------------------------------ arena
void Arena::reserve(size_t bytes) {
    if (bytes <= cap) return;
    if (base) HIP_CHECK(hipFree(base));
    base = nullptr;
    cap = 0;
    HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&base), bytes));
    cap = bytes;
    off = 0;
}
void* Arena::alloc(size_t bytes) {
    const size_t a = (off + 255) & ~size_t(255);
    if (a + bytes > cap) fail(BEVGEN_ERR_INTERNAL, "workspace arena exhausted: need %zu more bytes (capacity %zu)", a + bytes - cap, cap);
    off = a + bytes;
    if (off > high) high = off;
    return base + a;
}
void Arena::release() {
    if (base) (void)hipFree(base);
    base = nullptr;
    cap = off = 0;
}

// ------------------------------------------------------------------------------------------------ ctx
void Ctx::split_weight(const float* w, long n) {
    if (cfg.precision != BEVGEN_PRECISION_F16X3 || split.count(w)) return;
    void* planes = own((size_t)n * 4);
    const bool w16 = cfg.weight_dtype == BEVGEN_W_F16;
    // weight_dtype = f16: the model becomes the one whose matrices are f16-representable (rounded once, in place, so that every other consumer of
    // the fp32 copy sees the same values); the low plane is then zero and the GEMM issues two MFMAs per product instead of three
    if (w16) launch_round_to_f16(const_cast<float*>(w), nullptr, n, 0);
    launch_split_weight(w, planes, n, 0);
    split[w] = SplitPlanes{reinterpret_cast<const uint16_t*>(planes), reinterpret_cast<const uint16_t*>(planes) + 32, w16};
}

void Ctx::retire_graph(hipGraphExec_t e, hipGraph_t g) {
    // graphs retired by earlier calls have certainly been enqueued before this point; drain and free them
    if (!retired_graphs.empty() && graph_stream) {
        (void)hipStreamSynchronize(graph_stream);
        for (auto& p : retired_graphs) {
            (void)hipGraphExecDestroy(p.first);
            (void)hipGraphDestroy(p.second);
        }
        retired_graphs.clear();
    }
    retired_graphs.emplace_back(e, g);
}

Ctx::~Ctx() {
    if (graph_stream) (void)hipStreamSynchronize(graph_stream);
    for (auto& p : retired_graphs) {
        (void)hipGraphExecDestroy(p.first);
        (void)hipGraphDestroy(p.second);
    }
    if (graph_exec) { (void)hipGraphExecDestroy(graph_exec); (void)hipGraphDestroy(graph); }
    for (hipEvent_t e : step_events) (void)hipEventDestroy(e);
    if (graph_ev_in) (void)hipEventDestroy(graph_ev_in);
    if (graph_ev_out) (void)hipEventDestroy(graph_ev_out);
    if (graph_stream) (void)hipStreamDestroy(graph_stream);
    for (hipStream_t q : chain_streams) { (void)hipStreamSynchronize(q); (void)hipStreamDestroy(q); }
    if (chain_fork) (void)hipEventDestroy(chain_fork);
    for (hipEvent_t e : chain_join) (void)hipEventDestroy(e);
    for (auto& kv : params)
        if (kv.second.ptr) (void)hipFree(kv.second.ptr);
    for (void* p : owned) (void)hipFree(p);
    arena.release();
    persist.release();
    if (status_host) (void)hipHostFree(status_host);
}

// The host side of the device status word (common.h).  Called where the host is synchronised anyway (bevgen_synchronize, bevgen_finalize) and - without synchronising - at the
// entry of every C-ABI call, so that a flag raised by an earlier asynchronous call surfaces at the next one at the latest.  The word is cleared: one report per event.
void Ctx::check_status(const char* when) {
    if (!status_host) return;
    const unsigned e = __atomic_load_n(status_host, __ATOMIC_ACQUIRE);
    if (!e) return;
    __atomic_store_n(status_host, 0u, __ATOMIC_RELEASE);
    std::string m = std::string("device status word ") + std::to_string(e) + " (" + when + "): the results of the affected call are INVALID.";
    if (e & BG_ST_NONFINITE_LOGITS)
        m += "  A sampler read a NaN / inf logit or critic score (the reference asserts finite logits every step: mingpt_sparse.py:383,388, cond_transformer_multi_view.py:202).";
    if (e & BG_ST_F16_RANGE)
        m += "  A value written as an f16 operand (hi / lo planes of precision='f16x3', fp16 KV cache, fp16 decode activations) was NaN or had |v| >= 65520: the f16 split has a "
             "5-bit exponent where the reference's bf16 / fp32 arithmetic has 8.  Run this checkpoint with precision='fp32' (and kv_cache='f32').";
    if (e & BG_ST_NONFINITE_PIXELS) m += "  The VQGAN decoder produced a NaN / inf pixel.";
    if (e & (BG_ST_MLP_BARRIER | BG_ST_MLP_PLACEMENT)) {
        m += std::string("  The fused MLP launch of the decode step failed:") + ((e & BG_ST_MLP_BARRIER) ? " an XCD-local barrier timed out - the launch did not have the GPU to itself;" : "") +
             ((e & BG_ST_MLP_PLACEMENT) ? " a workgroup was not placed on the XCD its index implies;" : "") +
             " this context now runs the two-launch form without an in-kernel exchange - repeat the call ($BEVGEN_MLP_FUSE=0 selects that form from the start).";
        mlpf_disabled = true;
        if (graph_exec) { retire_graph(graph_exec, graph); graph_exec = nullptr; graph = nullptr; }   // the captured step contains the fused launch
    }
    fail((e & (BG_ST_NONFINITE_LOGITS | BG_ST_F16_RANGE | BG_ST_NONFINITE_PIXELS)) ? BEVGEN_ERR_NUMERIC : BEVGEN_ERR_INTERNAL, "%s", m.c_str());
}
const DevTensor* Ctx::find(const std::string& name) const {
    auto it = params.find(name);
    return it == params.end() ? nullptr : &it->second;
}
const DevTensor& Ctx::need(const std::string& name) const {
    auto it = params.find(name);
    if (it == params.end()) fail(BEVGEN_ERR_INVALID, "missing tensor '%s' (load it with bevgen_load_tensor before bevgen_finalize)", name.c_str());
    return it->second;
}
void* Ctx::own(size_t bytes) {
    void* p = nullptr;
    HIP_CHECK(hipMalloc(&p, bytes ? bytes : 16));
    owned.push_back(p);
    return p;
}

static size_t dtype_size(int dt) {
    switch (dt) {
        case BEVGEN_DTYPE_F32: return 4;
        case BEVGEN_DTYPE_I64: return 8;
        case BEVGEN_DTYPE_U8: return 1;
        case BEVGEN_DTYPE_F64: return 8;
    }
    fail(BEVGEN_ERR_INVALID, "unknown dtype %d", dt);
}

void ctx_load_tensor(Ctx& c, const char* name, const void* h, int dtype, int ndim, const int64_t* shape) {
    BG_REQUIRE(name && h && ndim >= 0 && ndim <= 8, "load_tensor: bad arguments");
    HIP_CHECK(hipSetDevice(c.device));
    DevTensor t;
    t.dtype = dtype;
    long n = 1;
    for (int i = 0; i < ndim; ++i) { t.shape.push_back(shape[i]); n *= shape[i]; }
    BG_REQUIRE(n > 0, "load_tensor('%s'): empty tensor", name);
    std::vector<float> conv;
    const void* src = h;
    if (dtype == BEVGEN_DTYPE_F64) {  // tables such as the legacy prob_matrix are float64 in the reference; arithmetic casts them to fp32
        conv.resize(n);
        const double* d = reinterpret_cast<const double*>(h);
        for (long i = 0; i < n; ++i) conv[i] = (float)d[i];
        src = conv.data();
        t.dtype = BEVGEN_DTYPE_F32;
    }
    t.bytes = (size_t)n * dtype_size(t.dtype);
    auto it = c.params.find(name);
    if (it != c.params.end()) {
        (void)hipFree(it->second.ptr);
        c.params.erase(it);
    }
    HIP_CHECK(hipMalloc(&t.ptr, t.bytes));
    HIP_CHECK(hipMemcpy(t.ptr, src, t.bytes, hipMemcpyHostToDevice));
    if (std::string(name) == "table.forward_shuffle_idx") {
        BG_REQUIRE(t.dtype == BEVGEN_DTYPE_I64, "table.forward_shuffle_idx must be int64");
        c.h_fwd_idx.assign(reinterpret_cast<const int64_t*>(h), reinterpret_cast<const int64_t*>(h) + n);
    }
    c.params[name] = std::move(t);
    c.finalized = false;
}

static void expect_shape(const Ctx& c, const std::string& name, std::initializer_list<int64_t> shape) {
    const DevTensor& t = c.need(name);
    long n = 1;
    for (auto d : shape) n *= d;
    if (t.numel() != n) {
        std::string got;
        for (auto d : t.shape) got += std::to_string(d) + ",";
        fail(BEVGEN_ERR_INVALID, "tensor '%s' has %ld elements (shape [%s]) but the configuration needs %ld", name.c_str(), t.numel(), got.c_str(), n);
    }
}

static void finalize_muse(Ctx& c) {
    const auto& g = c.cfg;
    const std::string p = "transformer.";
    const int D = c.D, H = c.H, F = c.F;
    BG_REQUIRE(H * 64 == D || true, "unused");
    expect_shape(c, p + "token_emb.weight", {g.vocab_size + 1, D});
    expect_shape(c, p + "pos_emb.weight", {c.N, D});
    expect_shape(c, p + "cond_token_emb.weight", {g.cond_vocab_size, D});
    expect_shape(c, p + "cond_pos_emb.weight", {c.K, D});
    expect_shape(c, p + "to_logits.weight", {g.vocab_size, D});
    expect_shape(c, p + "transformer_blocks.norm.gamma", {D});
    if (c.find("token_critic.to_pred.weight")) {   // absent for a MaskGit built without a token critic (scores then come from the softmax confidence, muse_net:611-622)
        expect_shape(c, "token_critic.to_pred.weight", {1, D});
        expect_shape(c, "token_critic.to_pred.bias", {1});
    }
    const int inner = H * 64;
    (void)xcd_placement_verified();   // (the device's one-time placement probe of the stream-K projections runs here, never on the sampling path)
    // the Route M workspace (qraw, att, split-K partial tiles) and every projection call size their rows by D: the released model has heads * 64 == dim
    BG_REQUIRE(inner == D, "Route M: num_heads * 64 = %d must equal dim = %d", inner, D);
    c.Fpad = (int)round_up(F, 32);
    c.muse.resize(g.num_layers);
    for (int i = 0; i < g.num_layers; ++i) {
        MuseLayer& l = c.muse[i];
        for (int j = 0; j < 2; ++j) {
            const std::string q = p + "transformer_blocks.layers." + std::to_string(i) + "." + std::to_string(j) + ".";
            expect_shape(c, q + "to_q.weight", {inner, D});
            expect_shape(c, q + "to_kv.weight", {2 * inner, D});
            expect_shape(c, q + "to_out.weight", {D, inner});
            expect_shape(c, q + "null_kv", {2, H, 1, 64});
            l.norm_g[j] = c.pf(q + "norm.gamma");
            l.to_q[j] = c.pf(q + "to_q.weight");
            l.to_kv[j] = c.pf(q + "to_kv.weight");
            l.to_out[j] = c.pf(q + "to_out.weight");
            l.q_scale[j] = c.pf(q + "q_scale");
            l.k_scale[j] = c.pf(q + "k_scale");
            l.null_kv[j] = c.pf(q + "null_kv");
        }
        const std::string q = p + "transformer_blocks.layers." + std::to_string(i) + ".2.";
        expect_shape(c, q + "1.weight", {2 * F, D});
        expect_shape(c, q + "4.weight", {D, F});
        expect_shape(c, q + "3.gamma", {F});
        l.ff_g0 = c.pf(q + "0.gamma");
        l.ff_w1 = c.pf(q + "1.weight");
        l.ff_g3 = c.pf(q + "3.gamma");
        l.ff_w4_padded = reinterpret_cast<float*>(c.own((size_t)D * c.Fpad * sizeof(float)));
        launch_pad_rows(c.pf(q + "4.weight"), F, l.ff_w4_padded, c.Fpad, D, F, 0);
        for (int j = 0; j < 2; ++j) {
            c.split_weight(l.to_q[j], (long)inner * D);
            c.split_weight(l.to_kv[j], 2L * inner * D);
            c.split_weight(l.to_out[j], (long)D * inner);
        }
        c.split_weight(l.ff_w1, 2L * F * D);
        if (g.precision == BEVGEN_PRECISION_F16X3) {
            l.null_self = c.own((size_t)4 * H * 64 * sizeof(_Float16));
            launch_muse_null_kv_prep(l.null_kv[0], l.k_scale[0], l.null_self, H, 0);
            // to_q | to_kv of the self-attention (muse_net:126-132: both read LayerNorm(x)) as one matrix: one projection per layer instead of two
            l.to_qkv_self = reinterpret_cast<float*>(c.own((size_t)3 * inner * D * sizeof(float)));
            HIP_CHECK(hipMemcpy(l.to_qkv_self, l.to_q[0], (size_t)inner * D * sizeof(float), hipMemcpyDeviceToDevice));
            HIP_CHECK(hipMemcpy(l.to_qkv_self + (size_t)inner * D, l.to_kv[0], (size_t)2 * inner * D * sizeof(float), hipMemcpyDeviceToDevice));
            c.split_weight(l.to_qkv_self, 3L * inner * D);
        }
        if (g.precision == BEVGEN_PRECISION_F16X3 && c.Fpad % 64 == 0) {
            l.ff_w1_geglu = reinterpret_cast<float*>(c.own((size_t)2 * c.Fpad * D * sizeof(float)));
            launch_geglu_weight_order(l.ff_w1, l.ff_w1_geglu, F, c.Fpad, D, 0);
            c.split_weight(l.ff_w1_geglu, 2L * c.Fpad * D);
        }
        c.split_weight(l.ff_w4_padded, (long)D * c.Fpad);
        if (g.precision == BEVGEN_PRECISION_F16X3 && g.weight_dtype != BEVGEN_W_F16 && l.ff_w1_geglu) {
            // consumer-side constants of the folded LayerNorms (muse.cpp muse_blocks; weight_dtype = f16 keeps the LayerNorm kernels: W o gamma is not f16-representable
            // and that mode's contract is "the fp32-class evaluation of the ROUNDED matrices")
            auto fold = [&](const float* W, const float* gamma, int N, int K, int Kg, float*& Wg, float*& cs) {
                Wg = reinterpret_cast<float*>(c.own((size_t)N * K * sizeof(float)));
                cs = reinterpret_cast<float*>(c.own((size_t)N * sizeof(float)));
                launch_ln_fold_weight(W, gamma, Wg, cs, N, K, Kg, 0);
                c.split_weight(Wg, (long)N * K);
            };
            fold(l.ff_w4_padded, l.ff_g3, D, c.Fpad, F, l.fold_w4, l.fold_w4_cs);
            fold(l.to_qkv_self, l.norm_g[0], 3 * inner, D, D, l.fold_qkv, l.fold_qkv_cs);
            fold(l.to_q[0], l.norm_g[0], inner, D, D, l.fold_q_self, l.fold_q_self_cs);
            fold(l.to_kv[0], l.norm_g[0], 2 * inner, D, D, l.fold_kv_self, l.fold_kv_self_cs);
            fold(l.to_q[1], l.norm_g[1], inner, D, D, l.fold_q_cross, l.fold_q_cross_cs);
            fold(l.ff_w1_geglu, l.ff_g0, 2 * c.Fpad, D, D, l.fold_w1, l.fold_w1_cs);
        }
    }
    c.split_weight(c.pf(p + "to_logits.weight"), (long)g.vocab_size * D);
    // attention bias matrices with the null-key column
    c.NkS_pad = (int)round_up(c.N + 1, 32);
    c.NkC_pad = (int)round_up(c.K + 1, 32);
    c.ldS = c.NkS_pad;
    c.ldC = c.NkC_pad;
    c.bias_self = reinterpret_cast<float*>(c.own((size_t)c.N * c.ldS * sizeof(float)));
    c.bias_cross = reinterpret_cast<float*>(c.own((size_t)c.N * c.ldC * sizeof(float)));
    BG_REQUIRE(c.L == c.N + c.K, "Route M requires num_pad_tokens == 0 (sparse_block_size 1): L=%d, N+K=%d (muse_maskgit_pytorch.py:152-154 slices the bias at K)", c.L, c.N + c.K);
    launch_build_muse_bias(c.attn_bias, c.L, c.K, c.N, c.bias_self, c.ldS, c.bias_cross, c.ldC, 0);
    if (g.precision == BEVGEN_PRECISION_F16X3) {  // the split-precision attention kernel evaluates the softmax as 2^x (kernels.h)
        launch_scale(c.bias_self, (long)c.N * c.ldS, kLog2e, 0);
        launch_scale(c.bias_cross, (long)c.N * c.ldC, kLog2e, 0);
        // ... and reads them as packed images: one wave load instruction = 1 KiB contiguous, in the accumulator order of its 32 x 32 score tile
        c.bias_self_pk = reinterpret_cast<float*>(c.own(attn_bias_packed_floats(c.N, c.NkS_pad) * sizeof(float)));
        c.bias_cross_pk = reinterpret_cast<float*>(c.own(attn_bias_packed_floats(c.N, c.NkC_pad) * sizeof(float)));
        launch_pack_attn_bias(c.bias_self, c.ldS, c.N, c.NkS_pad, c.bias_self_pk, 0);
        launch_pack_attn_bias(c.bias_cross, c.ldC, c.N, c.NkC_pad, c.bias_cross_pk, 0);
    }
}

// operand image of the fused QKV weight for the split decode layer's LayerNorm + projection kernel (packed like the MLP images)
static void pack_split_qkv_layer(Ctx& c, ArLayer& l, hipStream_t s) {
    const int D = c.D;
    const bool wf16 = c.cfg.decode_weight_dtype == BEVGEN_W_F16;
    l.wqkv_wp = reinterpret_cast<float*>(c.own(skinny_packed_floats(3 * D, D) * (wf16 ? sizeof(_Float16) : sizeof(float))));
    if (wf16) launch_pack_skinny_weight_f16(l.wqkv, l.wqkv_wp, 3 * D, D, s);
    else launch_pack_skinny_weight(l.wqkv, l.wqkv_wp, 3 * D, D, s);
}

static void finalize_ar(Ctx& c) {
    const auto& g = c.cfg;
    const int D = c.D;
    expect_shape(c, "x_tok_emb.weight", {g.vocab_size + 1, D});
    expect_shape(c, "cond_tok_emb.weight", {g.cond_vocab_size, D});
    expect_shape(c, "x_pos_emb", {1, c.N, D});
    expect_shape(c, "cond_pos_emb", {1, c.K, D});
    expect_shape(c, "head.weight", {g.vocab_size, D});
    const bool wf16 = g.decode_weight_dtype == BEVGEN_W_F16;
    const bool fused_like = g.decode_path == BEVGEN_DECODE_FUSED || g.decode_path == BEVGEN_DECODE_SPLIT || g.decode_path == BEVGEN_DECODE_AUTO;
    BG_REQUIRE(!wf16 || (fused_like && D % 256 == 0), "decode_weights = f16 needs the fused decode path and dim %% 256 == 0 (dim = %d)", D);
    BG_REQUIRE(g.weight_dtype != BEVGEN_W_F16 || wf16, "Route A: weight_dtype = f16 needs decode_weight_dtype = f16 as well (prefill and decode must run ONE rounded model)");
    c.ar.resize(g.num_layers);
    for (int i = 0; i < g.num_layers; ++i) {
        ArLayer& l = c.ar[i];
        const std::string q = "blocks." + std::to_string(i) + ".";
        for (const char* n : {"query", "key", "value"}) expect_shape(c, q + "attention." + n + ".weight", {D, D});
        expect_shape(c, q + "mlp.0.weight", {4 * D, D});
        expect_shape(c, q + "mlp.2.weight", {D, 4 * D});
        l.ln1_w = c.pf(q + "ln1.weight"); l.ln1_b = c.pf(q + "ln1.bias");
        l.ln2_w = c.pf(q + "ln2.weight"); l.ln2_b = c.pf(q + "ln2.bias");
        l.mlp0_w = c.pf(q + "mlp.0.weight"); l.mlp0_b = c.pf(q + "mlp.0.bias");
        l.mlp2_w = c.pf(q + "mlp.2.weight"); l.mlp2_b = c.pf(q + "mlp.2.bias");
        l.wqkv = reinterpret_cast<float*>(c.own((size_t)3 * D * D * sizeof(float)));
        l.bqkv = reinterpret_cast<float*>(c.own((size_t)3 * D * sizeof(float)));
        launch_fuse_qkv(c.pf(q + "attention.query.weight"), c.pf(q + "attention.key.weight"), c.pf(q + "attention.value.weight"),
                        c.pf(q + "attention.query.bias"), c.pf(q + "attention.key.bias"), c.pf(q + "attention.value.bias"), l.wqkv, l.bqkv, D, 0);
        if (wf16) {   // the model becomes the one whose projection weights are fp16-representable: every consumer below sees the rounded values
            l.wqkv_h = c.own((size_t)3 * D * D * sizeof(_Float16));
            launch_round_to_f16(l.wqkv, l.wqkv_h, 3L * D * D, 0);
            launch_round_to_f16(const_cast<float*>(l.mlp0_w), nullptr, 4L * D * D, 0);
            launch_round_to_f16(const_cast<float*>(l.mlp2_w), nullptr, 4L * D * D, 0);
        }
        if (fused_like) {
            // (after the rounding above: the constants belong to the matrix the decode step multiplies by)
            l.ln1_cs = reinterpret_cast<float*>(c.own((size_t)3 * D * sizeof(float)));
            l.ln1_ds = reinterpret_cast<float*>(c.own((size_t)3 * D * sizeof(float)));
            launch_ar_ln_fold(l.wqkv, l.bqkv, l.ln1_w, l.ln1_b, l.ln1_cs, l.ln1_ds, 3 * D, D, 0);
            l.mlp0_cs = reinterpret_cast<float*>(c.own((size_t)4 * D * sizeof(float)));
            l.mlp0_ds = reinterpret_cast<float*>(c.own((size_t)4 * D * sizeof(float)));
            launch_ar_ln_fold(l.mlp0_w, l.mlp0_b, l.ln2_w, l.ln2_b, l.mlp0_cs, l.mlp0_ds, 4 * D, D, 0);
            const size_t eb = wf16 ? sizeof(_Float16) : sizeof(float);
            // The split layer's QKV operand image (3 D D elements per layer, ~300 MB at config 4 in fp32) is packed NOW - at load time, where allocation failures
            // belong and no timed region is open - for decode_path = split and for auto unless the context was created for batches the auto rule never sends down
            // the split layer (cfg.max_batch > 4).  Only such a context, when it is later called with <= 4 sequences after all, packs on that first call
            // (ctx_pack_split_qkv: one allocation per layer + one stream synchronisation, once per context; documented in include/bevgen_hip.h)
            if (g.decode_path == BEVGEN_DECODE_SPLIT || (g.decode_path == BEVGEN_DECODE_AUTO && g.max_batch <= 4)) pack_split_qkv_layer(c, l, 0);
            l.mlp0_wp = reinterpret_cast<float*>(c.own(skinny_packed_floats(4 * D, D) * eb));
            l.mlp2_wp = reinterpret_cast<float*>(c.own(skinny_packed_floats(D, 4 * D) * eb));
            if (wf16) {
                launch_pack_skinny_weight_f16(l.mlp0_w, l.mlp0_wp, 4 * D, D, 0);
                launch_pack_skinny_weight_f16(l.mlp2_w, l.mlp2_wp, D, 4 * D, 0);
            } else {
                launch_pack_skinny_weight(l.mlp0_w, l.mlp0_wp, 4 * D, D, 0);
                launch_pack_skinny_weight(l.mlp2_w, l.mlp2_wp, D, 4 * D, 0);
            }
        }
        c.split_weight(l.wqkv, 3L * D * D);      // used by the prefill GEMMs (the per-token decode GEMMs stream the fp32 weights)
        c.split_weight(l.mlp0_w, 4L * D * D);
        c.split_weight(l.mlp2_w, 4L * D * D);
    }
    if (wf16) launch_round_to_f16(const_cast<float*>(c.pf("head.weight")), nullptr, (long)g.vocab_size * D, 0);
    if (fused_like) {   // state of the fused MLP launch (both projections of a layer in one launch, XCD-local exchange)
        c.mlpf_sync = reinterpret_cast<unsigned*>(c.own(mlp_fused_sync_words() * sizeof(unsigned)));
        HIP_CHECK(hipMemset(c.mlpf_sync, 0, mlp_fused_sync_words() * sizeof(unsigned)));
        (void)mlp_fused_supported(1, D, wf16);   // (runs the device's one-time placement probe here, never inside a stream capture)

    }
    if (fused_like) {
        c.head_wp = reinterpret_cast<float*>(c.own(skinny_packed_floats(g.vocab_size, D) * (wf16 ? sizeof(_Float16) : sizeof(float))));
        if (wf16) launch_pack_skinny_weight_f16(c.pf("head.weight"), c.head_wp, g.vocab_size, D, 0);
        else launch_pack_skinny_weight(c.pf("head.weight"), c.head_wp, g.vocab_size, D, 0);
    }
    // visibility mask: allowed AND layout block present.  The reference keeps one layout buffer PER LAYER in the checkpoint
    // (blocks.{i}.attention.sparse_self_attention.master_layout, drawn at construction when density < 1: mingpt_sparse.py:176, mask_generator.py:217-228);
    // a layer without one uses table.layout.  Identical layers / heads share one plane.
    const int blk = g.sparse_block_size, nb = c.L / blk;
    const size_t lay_n = (size_t)c.H * nb * nb;
    expect_shape(c, "table.attention_mask", {c.L, c.L});
    std::vector<const DevTensor*> lays(g.num_layers);
    std::vector<std::vector<int64_t>> hl(g.num_layers);
    for (int i = 0; i < g.num_layers; ++i) {
        const DevTensor* t = c.find("blocks." + std::to_string(i) + ".attention.sparse_self_attention.master_layout");
        if (!t) t = &c.need("table.layout");
        BG_REQUIRE(t->dtype == BEVGEN_DTYPE_I64 && (size_t)t->numel() == lay_n, "layout of layer %d must be int64 [H=%d, %d, %d]", i, c.H, nb, nb);
        lays[i] = t;
        hl[i].resize(lay_n);
        HIP_CHECK(hipMemcpy(hl[i].data(), t->ptr, t->bytes, hipMemcpyDeviceToHost));
    }
    bool same_layers = true, same_heads = true;
    for (int i = 0; i < g.num_layers; ++i) {
        if (i > 0 && hl[i] != hl[0]) same_layers = false;
        for (int h = 1; h < c.H && same_heads; ++h)
            same_heads = std::memcmp(hl[i].data(), hl[i].data() + (size_t)h * nb * nb, (size_t)nb * nb * sizeof(int64_t)) == 0;
    }
    c.keep_heads = same_heads ? 1 : c.H;
    c.keep_layers = same_layers ? 1 : g.num_layers;
    // can any row skip anything?  Rows are causal (keys beyond the row are never walked), so chunk lists only pay when a block at or below the diagonal is absent
    bool skippable = false;
    for (int i = 0; i < g.num_layers && !skippable; ++i)
        for (int h = 0; h < c.H && !skippable; ++h)
            for (int r = 0; r < nb && !skippable; ++r)
                for (int j = 0; j <= r && !skippable; ++j) skippable = hl[i][((size_t)h * nb + r) * nb + j] == 0;
    // does any layout hide an element the mask allows?  (density 1.0: no - the layout is the block cover of the mask - and the kernels then skip the layout row)
    bool lay_hides = false;
    {
        std::vector<float> hm((size_t)c.L * c.L);
        HIP_CHECK(hipMemcpy(hm.data(), c.pf("table.attention_mask"), hm.size() * sizeof(float), hipMemcpyDeviceToHost));
        std::vector<uint8_t> any((size_t)nb * nb, 0);   // block (rb, cb) holds an allowed element
        for (int r = 0; r < c.L; ++r)
            for (int k = 0; k < c.L; ++k)
                if (hm[(size_t)r * c.L + k] != 0.f) any[(size_t)(r / blk) * nb + k / blk] = 1;
        for (int i = 0; i < g.num_layers && !lay_hides; ++i)
            for (int h = 0; h < c.H && !lay_hides; ++h)
                for (size_t b = 0; b < (size_t)nb * nb && !lay_hides; ++b) lay_hides = any[b] && hl[i][(size_t)h * nb * nb + b] == 0;
    }
    c.lay_hides = lay_hides;
    c.allowed = reinterpret_cast<uint8_t*>(c.own((size_t)c.L * c.L));
    launch_build_allowed(c.pf("table.attention_mask"), c.allowed, (long)c.L * c.L, 0);
    const size_t lay_plane = (size_t)c.keep_heads * nb * nb;
    c.chunks_ld = (int)cdiv(c.L, 16) + 1;
    const size_t chunk_plane = (size_t)c.keep_heads * nb * c.chunks_ld;
    c.lay = reinterpret_cast<uint8_t*>(c.own(lay_plane * c.keep_layers));
    uint16_t* chunks = reinterpret_cast<uint16_t*>(c.own(chunk_plane * c.keep_layers * sizeof(uint16_t)));
    for (int i = 0; i < c.keep_layers; ++i)
        launch_build_layout(reinterpret_cast<const int64_t*>(lays[i]->ptr), c.lay + i * lay_plane, chunks + i * chunk_plane, c.keep_heads, nb, blk, c.L, c.chunks_ld, 0);
    // (a layout that hides an allowed element anywhere - also above the diagonal - makes vis_of_layer() hand out the layout: its chunk lists must go with it,
    // the fused decode kernel walks them: launch_ar_attn_fused requires lists whenever a layout is set)
    c.chunks = (skippable || lay_hides) ? chunks : nullptr;
    // prefill bias over the condition rows: scale*(bias) where visible, -1e30 elsewhere.  One image when the layers share their layout;
    // per-layer layouts (density < 1) build theirs in the prefill's workspace, layer by layer (ar.cpp)
    c.Kpad = (int)round_up(c.K, 32);
    c.prefill_bias = nullptr;
    if (c.keep_layers == 1) {
        c.prefill_bias = reinterpret_cast<float*>(c.own((size_t)c.keep_heads * c.K * c.Kpad * sizeof(float)));
        launch_build_masked_bias(c.attn_bias, c.vis_of_layer(0), c.prefill_bias, c.keep_heads, c.K, c.K, c.Kpad, c.L, 0.125f, 0);
    }
}

void ctx_pack_split_qkv(Ctx& c, hipStream_t s) {
    if (c.ar.empty() || c.ar[0].wqkv_wp) return;
    try {
        for (ArLayer& l : c.ar) pack_split_qkv_layer(c, l, s);
    } catch (const std::exception& e) {
        for (ArLayer& l : c.ar) l.wqkv_wp = nullptr;   // (buffers already taken stay in c.owned and are freed with the context)
        BG_REQUIRE(false, "decode_path = auto: packing the split decode layer's q/k/v operand image on the first call with <= 4 sequences failed (%s); create the context "
                          "with max_batch <= 4 or decode_path = split to take this memory at bevgen_finalize instead", e.what());
    }
    HIP_CHECK(hipStreamSynchronize(s));   // once per context, before the first split-path step is captured
}

void ctx_finalize(Ctx& c) {
    HIP_CHECK(hipSetDevice(c.device));
    const auto& g = c.cfg;
    // free previously derived buffers (re-finalize after reloading weights)
    if (c.graph_exec) {   // the captured graph bakes in pointers to the derived buffers freed below
        if (c.graph_stream) (void)hipStreamSynchronize(c.graph_stream);
        (void)hipGraphExecDestroy(c.graph_exec); (void)hipGraphDestroy(c.graph);
        c.graph_exec = nullptr; c.graph = nullptr;
    }
    for (void* p : c.owned) (void)hipFree(p);
    c.owned.clear();
    c.split.clear();
    c.T = g.cam_latent_h * g.cam_latent_w;
    c.N = c.T * g.num_cams;
    c.K = g.num_cond_tokens;
    c.L = g.seq_len;
    c.D = g.dim;
    c.H = g.num_heads;
    c.V = g.vocab_size;
    c.F = g.ff_inner;
    if (g.num_layers > 0) {
        BG_REQUIRE(c.D == c.H * 64, "head dimension must be 64 (dim=%d, heads=%d)", c.D, c.H);
        BG_REQUIRE(c.D % 32 == 0, "dim must be a multiple of 32");
        BG_REQUIRE(c.L >= c.N + c.K && c.L % g.sparse_block_size == 0, "seq_len %d inconsistent with N+K=%d / block %d", c.L, c.N + c.K, g.sparse_block_size);
        const std::string p = g.route == BEVGEN_ROUTE_MASKGIT ? "transformer." : "";
        if (g.image_embed) {
            expect_shape(c, "table.image_plane", {3, c.T});
            c.image_plane = c.pf("table.image_plane");
            expect_shape(c, p + "img_embed.weight", {c.D, 4});
            expect_shape(c, p + "cam_embed.weight", {c.D, 4});
        }
        if (g.bev_embed) {
            BG_REQUIRE(g.image_embed, "bev_embed requires image_embed (the camera origin embedding comes from cam_embed, mingpt_sparse.py:336-338)");
            expect_shape(c, p + "bev_embed.weight", {c.D, 2});
            expect_shape(c, p + "bev_cam_pos_emb", {1, g.num_cams, c.K, c.D});
            BG_REQUIRE(c.need(p + "bev_grid").numel() == 3L * c.K, "bev_grid must be [3, K]");
        }
        expect_shape(c, "table.forward_shuffle_idx", {c.N});
        c.fwd_idx = reinterpret_cast<const int64_t*>(c.need("table.forward_shuffle_idx").ptr);
        c.attn_bias = nullptr;
        if (g.camera_bias) {
            expect_shape(c, p + "camera_bias_emb", {(int64_t)c.L * (c.L + 1) / 2});
            expect_shape(c, "table.prob_matrix", {c.L, c.L});
            c.attn_bias = reinterpret_cast<float*>(c.own((size_t)c.L * c.L * sizeof(float)));
            launch_build_attn_bias(c.pf(p + "camera_bias_emb"), c.pf("table.prob_matrix"), c.attn_bias, c.L, 0);
        } else if (g.route == BEVGEN_ROUTE_MASKGIT) {
            c.attn_bias = reinterpret_cast<float*>(c.own((size_t)c.L * c.L * sizeof(float)));
            launch_build_attn_bias(nullptr, nullptr, c.attn_bias, c.L, 0);
        }
        if (g.route == BEVGEN_ROUTE_MASKGIT) finalize_muse(c);
        else finalize_ar(c);
    }
    if (g.vq_ch > 0) vq_finalize(c);
    HIP_CHECK(hipDeviceSynchronize());
    c.check_status("bevgen_finalize: a weight matrix");   // (a weight outside the f16 range of the split planes / the fp16 decode images)
    c.finalized = true;
}

}  // namespace bevgen
