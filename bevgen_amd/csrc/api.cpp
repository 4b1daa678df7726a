// extern "C" surface of libbevgen_hip (declared in include/bevgen_hip.h): argument validation, exception -> error-code
// translation.  No torch types cross this boundary: raw pointers, sizes and a hipStream_t.
#include <memory>

#include "model.h"
#include "profiler.h"

using namespace bevgen;

struct bevgen_ctx : public Ctx {};

static thread_local std::string g_create_error;

template <class F>
static int guarded(bevgen_ctx* ctx, F&& f, bool check_first = true) {
    try {
        if (!ctx) return BEVGEN_ERR_INVALID;
        HIP_CHECK(hipSetDevice(ctx->device));
        split_registry_set(ctx->cfg.precision == BEVGEN_PRECISION_F16X3 ? &ctx->split : nullptr);
        struct CurrentProf {   // this context's profiler and status word receive the launch records / flags of this call (restored on every exit path)
            Profiler* old;
            unsigned* old_st;
            CurrentProf(Profiler* p, unsigned* st) : old(prof_set_current(p)), old_st(status_set_current(st)) {}
            ~CurrentProf() { prof_set_current(old); status_set_current(old_st); }
        } cur(&ctx->prof, ctx->status_dev);
        // a flag raised by a kernel of an EARLIER (asynchronous) call: the host-visible word is read without synchronising, the error belongs to that call
        if (check_first) ctx->check_status("raised by an earlier call on this context");
        f();
        return BEVGEN_OK;
    } catch (const Error& e) {
        if (ctx) ctx->last_error = e.what();
        return e.code;
    } catch (const std::exception& e) {
        if (ctx) ctx->last_error = e.what();
        return BEVGEN_ERR_INTERNAL;
    }
}

static void need_final(bevgen_ctx* c) {
    if (!c->finalized) fail(BEVGEN_ERR_STATE, "bevgen_finalize must be called after loading tensors and before compute calls");
}

extern "C" {

int bevgen_abi_version(void) { return BEVGEN_ABI_VERSION; }

int bevgen_create(const bevgen_cfg* cfg, int device, bevgen_ctx** out) {
    try {
        BG_REQUIRE(cfg && out, "bevgen_create: null argument");
        BG_REQUIRE(cfg->abi_version == BEVGEN_ABI_VERSION, "bevgen_create: ABI version %d, library is %d", cfg->abi_version, BEVGEN_ABI_VERSION);
        BG_REQUIRE(cfg->route == BEVGEN_ROUTE_MASKGIT || cfg->route == BEVGEN_ROUTE_AR, "bevgen_create: unknown route %d", cfg->route);
        BG_REQUIRE(cfg->kv_cache_dtype == BEVGEN_KV_F32 || cfg->kv_cache_dtype == BEVGEN_KV_F16, "bevgen_create: kv_cache_dtype must be BEVGEN_KV_F32 or BEVGEN_KV_F16");
        BG_REQUIRE(cfg->precision == BEVGEN_PRECISION_FP32 || cfg->precision == BEVGEN_PRECISION_F16X3, "bevgen_create: precision must be BEVGEN_PRECISION_FP32 or BEVGEN_PRECISION_F16X3");
        BG_REQUIRE(cfg->weight_dtype == BEVGEN_W_F32 || (cfg->weight_dtype == BEVGEN_W_F16 && cfg->precision == BEVGEN_PRECISION_F16X3),
                   "bevgen_create: weight_dtype must be BEVGEN_W_F32, or BEVGEN_W_F16 together with BEVGEN_PRECISION_F16X3");
        int ndev = 0;
        HIP_CHECK(hipGetDeviceCount(&ndev));
        BG_REQUIRE(device >= 0 && device < ndev, "bevgen_create: device %d not present (%d visible)", device, ndev);
        hipDeviceProp_t prop;
        HIP_CHECK(hipGetDeviceProperties(&prop, device));
        BG_REQUIRE(std::string(prop.gcnArchName).rfind("gfx950", 0) == 0, "bevgen_create: device %d is %s; this library is built for gfx950 (MI355X) only", device,
                   prop.gcnArchName);
        HIP_CHECK(hipSetDevice(device));
        std::unique_ptr<bevgen_ctx> c(new bevgen_ctx());
        c->cfg = *cfg;
        c->device = device;
        // the device status word: mapped host memory, so that the host reads it without a copy or a synchronisation (common.h BG_ST_*)
        HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&c->status_host), 64, hipHostMallocMapped));
        *c->status_host = 0;
        HIP_CHECK(hipHostGetDevicePointer(reinterpret_cast<void**>(&c->status_dev), c->status_host, 0));
        *out = c.release();
        return BEVGEN_OK;
    } catch (const Error& e) {
        g_create_error = e.what();
        return e.code;
    } catch (const std::exception& e) {
        g_create_error = e.what();
        return BEVGEN_ERR_INTERNAL;
    }
}

void bevgen_destroy(bevgen_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    if (ctx->status_host && *ctx->status_host)   // nobody asked (no bevgen_synchronize, no later call): the last resort is to say it here
        fprintf(stderr, "libbevgen_hip: context destroyed with device status word %u pending (BEVGEN_STATUS_* in bevgen_hip.h): results of its last calls were invalid\n",
                *ctx->status_host);
    delete ctx;
}

const char* bevgen_last_error(const bevgen_ctx* ctx) { return ctx ? ctx->last_error.c_str() : g_create_error.c_str(); }

int bevgen_synchronize(bevgen_ctx* ctx, void* stream) {
    return guarded(ctx, [&] {
        HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
        if (ctx->graph_stream) HIP_CHECK(hipStreamSynchronize(ctx->graph_stream));   // (joined into `stream` by an event at the end of bevgen_ar_sample; belt and braces)
        ctx->check_status("bevgen_synchronize");
    }, false);
}

int bevgen_status(bevgen_ctx* ctx, unsigned* word) {
    if (!ctx || !word) return BEVGEN_ERR_INVALID;
    *word = ctx->status_host ? __atomic_load_n(ctx->status_host, __ATOMIC_ACQUIRE) : 0u;
    return BEVGEN_OK;
}

int bevgen_load_tensor(bevgen_ctx* ctx, const char* name, const void* h, int dtype, int ndim, const int64_t* shape) {
    return guarded(ctx, [&] { ctx_load_tensor(*ctx, name, h, dtype, ndim, shape); });
}

int bevgen_set_tables(bevgen_ctx* ctx, const int64_t* fwd, const float* mask, const int64_t* layout, const float* prob, const float* plane) {
    return guarded(ctx, [&] {
        const auto& g = ctx->cfg;
        const int64_t T = (int64_t)g.cam_latent_h * g.cam_latent_w, N = T * g.num_cams, L = g.seq_len, nb = L / g.sparse_block_size;
        if (fwd) { const int64_t s[1] = {N}; ctx_load_tensor(*ctx, "table.forward_shuffle_idx", fwd, BEVGEN_DTYPE_I64, 1, s); }
        if (mask) { const int64_t s[2] = {L, L}; ctx_load_tensor(*ctx, "table.attention_mask", mask, BEVGEN_DTYPE_F32, 2, s); }
        if (layout) { const int64_t s[3] = {g.num_heads, nb, nb}; ctx_load_tensor(*ctx, "table.layout", layout, BEVGEN_DTYPE_I64, 3, s); }
        if (prob) { const int64_t s[2] = {L, L}; ctx_load_tensor(*ctx, "table.prob_matrix", prob, BEVGEN_DTYPE_F32, 2, s); }
        if (plane) { const int64_t s[2] = {3, T}; ctx_load_tensor(*ctx, "table.image_plane", plane, BEVGEN_DTYPE_F32, 2, s); }
    });
}

int bevgen_finalize(bevgen_ctx* ctx) {
    return guarded(ctx, [&] { ctx_finalize(*ctx); });
}

int bevgen_muse_forward(bevgen_ctx* ctx, const int64_t* ids, const int64_t* cond, const float* I_inv, const float* E_inv, int B, float* logits, float* embed, void* stream) {
    return guarded(ctx, [&] {
        need_final(ctx);
        BG_REQUIRE(ids && cond && I_inv && E_inv, "muse_forward: null input");
        BG_REQUIRE(B >= 1 && (ctx->cfg.max_batch <= 0 || B <= ctx->cfg.max_batch), "muse_forward: batch %d exceeds max_batch %d", B, ctx->cfg.max_batch);
        muse_forward(*ctx, ids, cond, I_inv, E_inv, B, logits, embed, (hipStream_t)stream);
    });
}

int bevgen_maskgit_generate(bevgen_ctx* ctx, const int64_t* cond, const float* I_inv, const float* E_inv, int B, int timesteps, const int32_t* sched, float temperature,
                            int topk_k, float critic_noise_scale, const float* gumbel_u, const float* critic_u, const int64_t* init_ids, int64_t* out,
                            unsigned long long noise_seed, void* stream) {
    return bevgen_maskgit_generate_ex(ctx, cond, I_inv, E_inv, B, timesteps, sched, temperature, topk_k, critic_noise_scale, gumbel_u, critic_u, init_ids, out, noise_seed, 0, 1, stream);
}

int bevgen_maskgit_generate_ex(bevgen_ctx* ctx, const int64_t* cond, const float* I_inv, const float* E_inv, int B, int timesteps, const int32_t* sched, float temperature,
                               int topk_k, float critic_noise_scale, const float* gumbel_u, const float* critic_u, const int64_t* init_ids, int64_t* out,
                               unsigned long long noise_seed, int score_mode, int samples_per_layout, void* stream) {
    return guarded(ctx, [&] {
        need_final(ctx);
        BG_REQUIRE(cond && I_inv && E_inv && out && sched, "maskgit_generate: null argument");
        BG_REQUIRE(B >= 1 && (ctx->cfg.max_batch <= 0 || B <= ctx->cfg.max_batch), "maskgit_generate: batch %d exceeds max_batch %d", B, ctx->cfg.max_batch);
        BG_REQUIRE(topk_k >= 1 && topk_k <= ctx->cfg.vocab_size, "maskgit_generate: topk_k=%d out of range", topk_k);
        for (int i = 0; i < timesteps; ++i)
            BG_REQUIRE(sched[i] >= 1 && sched[i] <= ctx->cfg.cam_latent_h * ctx->cfg.cam_latent_w, "maskgit_generate: mask_schedule[%d]=%d out of range", i, sched[i]);
        BG_REQUIRE(samples_per_layout >= 1 && B % samples_per_layout == 0, "maskgit_generate: batch %d is not a multiple of samples_per_layout %d", B, samples_per_layout);
        maskgit_generate(*ctx, cond, I_inv, E_inv, B, timesteps, sched, temperature, topk_k, critic_noise_scale, gumbel_u, critic_u, init_ids, out, (hipStream_t)stream, noise_seed,
                         score_mode, samples_per_layout);
    });
}

int bevgen_op_philox_uniform(bevgen_ctx* ctx, unsigned long long seed, unsigned iter, unsigned stream_id, int V, long n, float* out, void* stream) {
    return guarded(ctx, [&] {
        BG_REQUIRE(out && n >= 0, "op_philox_uniform: bad arguments");
        launch_philox_fill(out, n, seed, iter, stream_id, V, (hipStream_t)stream);
    });
}

int bevgen_sparse_self_attention(bevgen_ctx* ctx, const float* q, const float* k, const float* v, const int64_t* layout, const float* mask, const float* add, int B,
                                 int H, int L, int block, float* out, void* stream) {
    return guarded(ctx, [&] {
        BG_REQUIRE(q && k && v && layout && mask && out, "sparse_self_attention: null argument");
        BG_REQUIRE(B >= 1 && H >= 1 && L >= 1 && block >= 1, "sparse_self_attention: bad sizes");
        sparse_self_attention_op(*ctx, q, k, v, layout, mask, add, B, H, L, block, out, (hipStream_t)stream);
    });
}

int bevgen_ar_prefill(bevgen_ctx* ctx, const int64_t* cond, const float* I_inv, const float* E_inv, int B, void* stream) {
    return guarded(ctx, [&] {
        need_final(ctx);
        BG_REQUIRE(cond && I_inv && E_inv, "ar_prefill: null input");
        ar_prefill(*ctx, cond, I_inv, E_inv, B, (hipStream_t)stream);
    });
}

int bevgen_ar_logits(bevgen_ctx* ctx, float* logits, void* stream) {
    return guarded(ctx, [&] {
        need_final(ctx);
        BG_REQUIRE(logits, "ar_logits: null output");
        ar_logits(*ctx, logits, (hipStream_t)stream);
    });
}

int bevgen_ar_decode_step(bevgen_ctx* ctx, const int64_t* tok, void* stream) {
    return guarded(ctx, [&] {
        need_final(ctx);
        BG_REQUIRE(tok, "ar_decode_step: null token pointer");
        ar_decode_step(*ctx, tok, (hipStream_t)stream);
    });
}

int bevgen_ar_sample(bevgen_ctx* ctx, const int64_t* cond, const float* I_inv, const float* E_inv, int B, int steps, int top_k, float temperature, int greedy,
                     const float* noise_u, int samples_per_layout, int64_t* out, float* step_logits, void* stream) {
    return guarded(ctx, [&] {
        need_final(ctx);
        BG_REQUIRE(cond && I_inv && E_inv && out, "ar_sample: null argument");
        BG_REQUIRE(temperature > 0.f, "ar_sample: temperature must be positive");
        ar_sample(*ctx, cond, I_inv, E_inv, B, steps, top_k, temperature, greedy, noise_u, samples_per_layout, nullptr, out, step_logits, (hipStream_t)stream);
    });
}

int bevgen_ar_sample_forced(bevgen_ctx* ctx, const int64_t* cond, const float* I_inv, const float* E_inv, int B, int steps, int top_k, float temperature, int greedy,
                            const float* noise_u, int samples_per_layout, const int64_t* forced, int64_t* out, float* step_logits, void* stream) {
    return guarded(ctx, [&] {
        need_final(ctx);
        BG_REQUIRE(cond && I_inv && E_inv && out, "ar_sample_forced: null argument");
        BG_REQUIRE(temperature > 0.f, "ar_sample_forced: temperature must be positive");
        ar_sample(*ctx, cond, I_inv, E_inv, B, steps, top_k, temperature, greedy, noise_u, samples_per_layout, forced, out, step_logits, (hipStream_t)stream);
    });
}

static void vq_default_grid(const bevgen_ctx* ctx, int& h, int& w) {   // 0 x 0 = the square grid of ddconfig.resolution
    if (h <= 0 || w <= 0) h = w = ctx->cfg.vq_resolution >> (ctx->cfg.vq_num_levels - 1);
}

int bevgen_vq_decode(bevgen_ctx* ctx, const int64_t* ids, int n, int lat_h, int lat_w, int out_mode, void* out, void* stream) {
    return guarded(ctx, [&] {
        need_final(ctx);
        BG_REQUIRE(ids && out && n >= 1 && out_mode >= 0 && out_mode <= 2, "vq_decode: bad arguments");
        vq_default_grid(ctx, lat_h, lat_w);
        vq_decode(*ctx, ids, nullptr, n, lat_h, lat_w, out_mode, out, (hipStream_t)stream);
    });
}

int bevgen_vq_encode(bevgen_ctx* ctx, const float* x, int n, int H, int W, int64_t* ids, void* stream) {
    return guarded(ctx, [&] {
        need_final(ctx);
        BG_REQUIRE(x && ids && n >= 1, "vq_encode: bad arguments");
        if (H <= 0 || W <= 0) H = W = ctx->cfg.vq_resolution;
        vq_encode(*ctx, x, n, H, W, ids, (hipStream_t)stream);
    });
}

int bevgen_vq_decode_latents(bevgen_ctx* ctx, const float* zq, int n, int lat_h, int lat_w, int out_mode, void* out, void* stream) {
    return guarded(ctx, [&] {
        need_final(ctx);
        BG_REQUIRE(zq && out && n >= 1 && out_mode >= 0 && out_mode <= 2, "vq_decode_latents: bad arguments");
        vq_default_grid(ctx, lat_h, lat_w);
        vq_decode(*ctx, nullptr, zq, n, lat_h, lat_w, out_mode, out, (hipStream_t)stream);
    });
}

int bevgen_op_gemm(bevgen_ctx* ctx, const float* a, const float* w, const float* bias, const float* residual, float* c, int M, int N, int K, int act_gelu, int skinny,
                   void* stream) {
    const int skinny_in = skinny;
    return guarded(ctx, [&] {
        GemmArgs g;
        g.A = a; g.B = w; g.C = c; g.R = residual; g.bias_n = bias;
        g.M = M; g.N = N; g.K = K; g.lda = K; g.ldb = K; g.ldc = N; g.ldr = N;
        g.act = act_gelu ? ACT_GELU : ACT_NONE;
        if (skinny == 5 || skinny == 6) skinny = 3;   // 5 = mode 3 with the k range split over three slices (the low-latency path of small batches); needs M small enough for
        const bool ksplit3 = skinny_in == 5;           // 128-row tiles.  6 = mode 3 in its stream-K form (gemm_split_glds_sk_kernel), whatever the shape
        const bool stream_k = skinny_in == 6;
        if (skinny == 4 || skinny == 3 || skinny == 2) {  // split-precision paths with on-the-fly operand splits (tests / roofline probes): 3 = LDS-DMA kernel, 4 = the same for
                                                          // an f16-representable w (two MFMAs per product: the weights='f16' mode's kernel)
            ctx->arena.reserve(((size_t)N * K + (size_t)M * K) * 4 + (ksplit3 ? (size_t)3 * M * N * 4 : 0) + (stream_k ? gemm_sk_ws_bytes() : 0) + 8192);
            ctx->arena.reset();
            uint16_t* bp = reinterpret_cast<uint16_t*>(ctx->arena.alloc((size_t)N * K * 4));
            launch_split_weight(w, bp, (long)N * K, (hipStream_t)stream);
            g.B_hi = bp; g.B_lo = bp + 32;
            if (skinny >= 3) {
                g.b_lo_zero = skinny == 4;
                uint16_t* ap = reinterpret_cast<uint16_t*>(ctx->arena.alloc((size_t)M * K * 4));
                launch_split_weight(a, ap, (long)M * K, (hipStream_t)stream);
                g.A_hi = ap; g.A_lo = ap + 32;
                if (ksplit3) { g.ksplit = 3; g.kpart = ctx->arena.get<float>((size_t)3 * M * N); }
                if (stream_k) {
                    g.sk_ws = ctx->arena.alloc(gemm_sk_ws_bytes());
                    HIP_CHECK(hipMemsetAsync(g.sk_ws, 0, 1024 * sizeof(unsigned), (hipStream_t)stream));
                    g.sk_epoch = 1; g.sk_force = true;
                }
                launch_gemm_split_glds(g, (hipStream_t)stream);
            } else {
                launch_gemm_split(g, (hipStream_t)stream);
            }
        } else if (skinny) {   // split-K workspace from this context's arena (one per context / device)
            ctx->arena.reserve(gemm_skinny_ws_bytes(M, N, K) + 4096);
            ctx->arena.reset();
            launch_gemm_skinny_ws(g, reinterpret_cast<float*>(ctx->arena.alloc(gemm_skinny_ws_bytes(M, N, K))), (hipStream_t)stream);
        }
        else launch_gemm(g, (hipStream_t)stream);
    });
}

int bevgen_op_ln_gemm(bevgen_ctx* ctx, const float* a, const float* ln_w, const float* ln_b, float eps, const float* w, const float* bias, float* c, int M, int N, int K,
                      int act_gelu, int ksplit, int* ksplit_out, void* stream) {
    return guarded(ctx, [&] {
        BG_REQUIRE(skinny_fused_supported(M, N, K, ln_w != nullptr), "op_ln_gemm: unsupported shape M=%d N=%d K=%d", M, N, K);
        SkinnyFusedArgs g;
        g.A = a; g.lda = K; g.ln_w = ln_w; g.ln_b = ln_b; g.eps = eps;
        ctx->arena.reserve((skinny_packed_floats(N, K) + 2 * (size_t)N) * sizeof(float) + 8192);
        ctx->arena.reset();
        float* wp = ctx->arena.get<float>(skinny_packed_floats(N, K));
        launch_pack_skinny_weight(w, wp, N, K, (hipStream_t)stream);
        g.Wp = wp; g.bias = bias; g.C = c; g.ldc = N;
        g.M = M; g.N = N; g.K = K; g.act = act_gelu ? ACT_GELU : ACT_NONE;
        g.ksplit = ksplit > 0 ? ksplit : (ln_w ? 1 : skinny_fused_ksplit(N, K));
        if (ksplit == -1) {   // the form the decode step launches for ln2 + MLP-up: LayerNorm folded into the product, row constants from launch_ar_ln_fold
            BG_REQUIRE(ln_w && ln_b && bias, "op_ln_gemm: ksplit = -1 (folded LayerNorm) needs gamma, beta and a bias");
            float* cs = ctx->arena.get<float>((size_t)N);
            float* ds = ctx->arena.get<float>((size_t)N);
            launch_ar_ln_fold(w, bias, ln_w, ln_b, cs, ds, N, K, (hipStream_t)stream);
            g.ln_cs = cs; g.ln_ds = ds;
        }
        if (ksplit_out) *ksplit_out = g.ksplit;
        g.trace = ctx->trace ? ctx->trace + 4096 * 8 : nullptr;
        launch_skinny_fused(g, (hipStream_t)stream);
    });
}

int bevgen_op_mlp_fused(bevgen_ctx* ctx, const float* x, const float* ln_w, const float* ln_b, float eps, const float* w1, const float* b1, const float* w2, const float* b2,
                        int w_f16, float* out, int M, int D, void* stream) {
    return guarded(ctx, [&] {
        hipStream_t s = (hipStream_t)stream;
        BG_REQUIRE(x && ln_w && ln_b && w1 && b1 && w2 && b2 && out, "op_mlp_fused: missing operand");
        BG_REQUIRE(mlp_fused_supported(M, D, w_f16 != 0), "op_mlp_fused: unsupported shape M=%d D=%d (or a device with fewer CUs than workgroups / BEVGEN_MLP_FUSE=0)", M, D);
        const size_t eb = w_f16 ? sizeof(_Float16) : sizeof(float);
        const size_t wbytes = (skinny_packed_floats(4 * D, D) + skinny_packed_floats(D, 4 * D)) * eb;
        const size_t fl = (size_t)8 * D + (size_t)M * 4 * D + (size_t)MLP_FUSED_PLANES * M * D + (size_t)M * D + 2 * (size_t)4 * D * D;
        ctx->arena.reserve(wbytes + fl * sizeof(float) + mlp_fused_sync_words() * sizeof(unsigned) + 64 * 256);
        ctx->arena.reset();
        void* w1p = ctx->arena.alloc(skinny_packed_floats(4 * D, D) * eb);
        void* w2p = ctx->arena.alloc(skinny_packed_floats(D, 4 * D) * eb);
        float* cs = ctx->arena.get<float>((size_t)4 * D);
        float* ds = ctx->arena.get<float>((size_t)4 * D);
        float* hidden = ctx->arena.get<float>((size_t)M * 4 * D);
        float* part = ctx->arena.get<float>((size_t)MLP_FUSED_PLANES * M * D);
        float* zero = ctx->arena.get<float>((size_t)M * D);
        unsigned* sync = ctx->arena.get<unsigned>(mlp_fused_sync_words());
        const float *w1u = w1, *w2u = w2;
        if (w_f16) {   // the model decode_weights = f16 defines: matrices rounded to fp16-representable values (row constants from the rounded matrix, as bevgen_finalize does)
            float* w1r = ctx->arena.get<float>((size_t)4 * D * D);
            float* w2r = ctx->arena.get<float>((size_t)4 * D * D);
            HIP_CHECK(hipMemcpyAsync(w1r, w1, (size_t)4 * D * D * sizeof(float), hipMemcpyDeviceToDevice, s));
            HIP_CHECK(hipMemcpyAsync(w2r, w2, (size_t)4 * D * D * sizeof(float), hipMemcpyDeviceToDevice, s));
            launch_round_to_f16(w1r, nullptr, (long)4 * D * D, s);
            launch_round_to_f16(w2r, nullptr, (long)4 * D * D, s);
            w1u = w1r; w2u = w2r;
            launch_pack_skinny_weight_f16(w1u, w1p, 4 * D, D, s);
            launch_pack_skinny_weight_f16(w2u, w2p, D, 4 * D, s);
        } else {
            launch_pack_skinny_weight(w1u, reinterpret_cast<float*>(w1p), 4 * D, D, s);
            launch_pack_skinny_weight(w2u, reinterpret_cast<float*>(w2p), D, 4 * D, s);
        }
        launch_ar_ln_fold(w1u, b1, ln_w, ln_b, cs, ds, 4 * D, D, s);
        HIP_CHECK(hipMemsetAsync(zero, 0, (size_t)M * D * sizeof(float), s));
        HIP_CHECK(hipMemsetAsync(sync, 0, mlp_fused_sync_words() * sizeof(unsigned), s));
        MlpFusedArgs g;
        g.A = x; g.lda = D; g.ln_w = ln_w; g.ln_cs = cs; g.ln_ds = ds; g.eps = eps;
        g.Wup = reinterpret_cast<const float*>(w1p); g.Wdn = reinterpret_cast<const float*>(w2p); g.w_f16 = w_f16;
        g.hidden = hidden; g.C = part; g.sync = sync; g.err = ctx->status_dev; g.M = M; g.D = D;
        g.trace = ctx->trace ? ctx->trace + 4096 * 8 : nullptr;
        // (twice: the second launch runs on the barrier state the first one left behind - the self-cleaning property the hipGraph replay of the decode step relies on)
        launch_ar_mlp_fused(g, s);
        launch_ar_mlp_fused(g, s);
        RowSrc r;
        r.base = zero; r.ld = D; r.partial = part; r.ns = MLP_FUSED_PLANES; r.pstride = (long)M * D; r.pld = D; r.bias = b2;
        launch_rowsrc_materialize(r, out, M, D, nullptr, s);
        HIP_CHECK(hipStreamSynchronize(s));
        ctx->check_status("op_mlp_fused");
    });
}

int bevgen_op_layernorm(bevgen_ctx* ctx, const float* x, const float* gamma, const float* beta, float* y, int rows, int D, float eps, void* stream) {
    return guarded(ctx, [&] { launch_layernorm(x, D, gamma, beta, y, D, rows, D, eps, (hipStream_t)stream); });
}

int bevgen_op_geglu_layernorm(bevgen_ctx* ctx, const float* h, const float* gamma, float* y, int rows, int F, int ldy, void* stream) {
    return guarded(ctx, [&] { launch_geglu_layernorm(h, 2 * F, gamma, y, ldy, rows, F, 1e-5f, (hipStream_t)stream); });
}

int bevgen_op_attention(bevgen_ctx* ctx, const float* q, const float* k, const float* v, const float* bias, int ldbias, int B, int H, int Nq, int Nk_pad, float scale,
                        float* out, void* stream) {
    return bevgen_op_attention_ex(ctx, q, k, v, bias, ldbias, B, H, Nq, Nk_pad, scale, 1, out, stream);
}

int bevgen_op_attention_ex(bevgen_ctx* ctx, const float* q, const float* k, const float* v, const float* bias, int ldbias, int B, int H, int Nq, int Nk_pad, float scale,
                           int key_splits, float* out, void* stream) {
    return guarded(ctx, [&] {
        if (ctx->cfg.precision == BEVGEN_PRECISION_F16X3) {
            // the split-precision flash-attention kernel of Route M on caller-supplied fp32 operands: operand images and the packed bias are built here
            BG_REQUIRE(Nk_pad % 32 == 0 && (bias == nullptr || (ldbias % 4 == 0 && ldbias >= Nk_pad)), "op_attention (split precision): Nk_pad %% 32, bias row stride %% 4 and >= Nk_pad");
            hipStream_t s = (hipStream_t)stream;
            const size_t nq = (size_t)B * H * Nq * 64, nk = (size_t)B * H * Nk_pad * 64;
            const size_t bpk = bias ? (size_t)attn_bias_packed_floats(Nq, Nk_pad) : 0, braw = bias ? (size_t)Nq * ldbias : 0;
            BG_REQUIRE(key_splits >= 1 && key_splits <= 8 && Nk_pad / 32 >= key_splits, "op_attention: key_splits=%d out of range for Nk_pad=%d", key_splits, Nk_pad);
            const size_t kws = key_splits > 1 ? (size_t)attn_split_ws_floats(B, H, Nq, key_splits) : 0;
            ctx->arena.reserve((nq + 2 * nk) * 4 + (bpk + braw + kws) * 4 + 8192);
            ctx->arena.reset();
            _Float16* Qp = reinterpret_cast<_Float16*>(ctx->arena.alloc(nq * 4));
            _Float16* Kp = reinterpret_cast<_Float16*>(ctx->arena.alloc(nk * 4));
            _Float16* Vp = reinterpret_cast<_Float16*>(ctx->arena.alloc(nk * 4));
            launch_attn_split_operands(q, k, v, Qp, Qp + nq, Kp, Kp + nk, Vp, Vp + nk, B, H, Nq, Nk_pad, scale * kLog2e, s);
            AttnSplitArgs sa{};
            sa.Qh = Qp; sa.Ql = Qp + nq; sa.Kh = Kp; sa.Kl = Kp + nk; sa.VTh = Vp; sa.VTl = Vp + nk;
            if (bias) {
                float* b2 = reinterpret_cast<float*>(ctx->arena.alloc(braw * 4));
                float* pk = reinterpret_cast<float*>(ctx->arena.alloc(bpk * 4));
                HIP_CHECK(hipMemcpyAsync(b2, bias, braw * 4, hipMemcpyDeviceToDevice, s));
                launch_scale(b2, (long)braw, kLog2e, s);
                launch_pack_attn_bias(b2, ldbias, Nq, Nk_pad, pk, s);
                sa.bias = b2; sa.bias_pk = pk; sa.ldbias = ldbias;
            }
            sa.bias_head_stride = 0;
            if (key_splits > 1) { sa.ksplit = key_splits; sa.kws = reinterpret_cast<float*>(ctx->arena.alloc(kws * 4)); }
            sa.O = out; sa.Op = nullptr; sa.B = B; sa.H = H; sa.Nq = Nq; sa.Nk_pad = Nk_pad; sa.scale = scale * kLog2e;
            sa.o_bstride = (long)Nq * H * 64; sa.o_qstride = (long)H * 64; sa.o_hstride = 64;
            launch_attention_split(sa, s);
            return;
        }
        AttnArgs a{};
        a.Q = q; a.K = k; a.V = v; a.bias = bias; a.R = nullptr; a.O = out;
        a.B = B; a.H = H; a.Nq = Nq; a.Nk_pad = Nk_pad;
        a.q_bstride = (long)H * Nq * 64; a.q_hstride = (long)Nq * 64;
        a.kv_bstride = (long)H * Nk_pad * 64; a.kv_hstride = (long)Nk_pad * 64;
        a.ldbias = ldbias; a.bias_head_stride = 0; a.scale = scale;
        a.o_bstride = (long)Nq * H * 64; a.o_qstride = (long)H * 64; a.o_hstride = 64;
        launch_attention(a, (hipStream_t)stream);
    });
}

int bevgen_op_decode_attention(bevgen_ctx* ctx, const float* q, const void* kc, const void* vc, int kv_dtype, const float* bias, int ldbias, const uint8_t* keep,
                               int ldkeep, long keep_head_stride, int B, int H, int n, int Lmax, float scale, float* out, void* stream) {
    return guarded(ctx, [&] {
        DecodeAttnArgs a;
        a.q = q; a.ldq = H * 64; a.kcache = kc; a.vcache = vc; a.kv_dtype = kv_dtype;
        a.bias = bias; a.ldbias = ldbias;
        a.vis.allowed = keep; a.vis.ldallowed = ldkeep; a.vis.allowed_head_stride = keep_head_stride;   // a dense per-head plane is an element mask with a head stride
        a.O = out; a.ldo = H * 64; a.B = B; a.H = H; a.n = n; a.Lmax = Lmax; a.scale = scale;
        const int S = decode_attention_splits(a.B, a.H, a.n);
        ctx->arena.reserve(decode_attention_ws_bytes(a.B, a.H, S) + 4096);
        ctx->arena.reset();
        launch_decode_attention_ws(a, reinterpret_cast<float*>(ctx->arena.alloc(decode_attention_ws_bytes(a.B, a.H, S))), S, (hipStream_t)stream);
    });
}

int bevgen_op_ar_attn_fused(bevgen_ctx* ctx, const float* x, const float* partial, int ns, const float* rbias, const float* ln_w, const float* ln_b, const float* wqkv,
                            const float* bqkv, int w_f16, void* kc, void* vc, int kv_dtype, const float* bias, int ldbias, const float* attn_mask, const int64_t* layout,
                            int block, int B, int G, int H, int n, int Lmax, int prefix, int split, float* out, void* stream) {
    return guarded(ctx, [&] {
        hipStream_t s = (hipStream_t)stream;
        const int D = H * 64, L = Lmax;
        BG_REQUIRE(x && ln_w && ln_b && wqkv && bqkv && kc && vc && out && n >= 1 && n <= Lmax, "op_ar_attn_fused: bad arguments");
        BG_REQUIRE(ar_attn_fused_supported(B, G, D, H), "op_ar_attn_fused: unsupported shape B=%d G=%d D=%d H=%d", B, G, D, H);
        BG_REQUIRE(!layout || (block >= 1 && L % block == 0), "op_ar_attn_fused: Lmax %d is not a multiple of the block size %d", L, block);
        const int nb = layout ? L / block : 0, cld = (int)cdiv(L, 16) + 1;
        ctx->arena.reserve((size_t)L * L + (size_t)H * nb * nb + (size_t)H * nb * cld * 2 + (size_t)3 * D * D * 6 + (size_t)6 * D * 4 + (split ? skinny_packed_floats(3 * D, D) * 4 + (size_t)B * 4 * D * 4 + (size_t)B * H * (split > 1 ? split : 0) * 66 * 4 : 0) + 14 * 256);
        ctx->arena.reset();
        ArAttnFusedArgs a;
        a.x.base = x; a.x.ld = D;
        if (partial && ns > 0) { a.x.partial = partial; a.x.ns = ns; a.x.pstride = (long)B * D; a.x.pld = D; }
        a.x.bias = rbias;
        a.ln_w = ln_w; a.ln_b = ln_b; a.wqkv = wqkv; a.bqkv = bqkv;
        const float* w_rounded = nullptr;
        if (w_f16) {
            void* h = ctx->arena.alloc((size_t)3 * D * D * 2);
            float* tmp = ctx->arena.get<float>((size_t)3 * D * D);   // round_to_f16 rounds in place: work on a copy of the caller's matrix (arena: no allocation, no sync)
            HIP_CHECK(hipMemcpyAsync(tmp, wqkv, (size_t)3 * D * D * 4, hipMemcpyDeviceToDevice, s));
            launch_round_to_f16(tmp, h, 3L * D * D, s);
            a.wqkv_h = h;
            w_rounded = tmp;
        }
        {   // row constants of the folded ln1, on the matrix the kernel multiplies by (the fp16-rounded one with w_f16)
            float* cs = ctx->arena.get<float>((size_t)3 * D);
            float* ds = ctx->arena.get<float>((size_t)3 * D);
            launch_ar_ln_fold(w_f16 ? w_rounded : wqkv, bqkv, ln_w, ln_b, cs, ds, 3 * D, D, s);
            a.ln_cs = cs; a.ln_ds = ds;
        }
        a.kcache = kc; a.vcache = vc; a.kv_dtype = kv_dtype;
        a.bias = bias; a.ldbias = ldbias;
        if (attn_mask) {
            uint8_t* al = ctx->arena.get<uint8_t>((size_t)L * L);
            launch_build_allowed(attn_mask, al, (long)L * L, s);
            a.vis.allowed = al; a.vis.ldallowed = L;
        }
        if (layout) {
            uint8_t* lay = ctx->arena.get<uint8_t>((size_t)H * nb * nb);
            uint16_t* ch = ctx->arena.get<uint16_t>((size_t)H * nb * cld);
            launch_build_layout(layout, lay, ch, H, nb, block, L, cld, s);
            a.vis.lay = lay; a.vis.lay_head_stride = (long)nb * nb; a.vis.nb = nb; a.vis.blk = block;
            a.vis.chunks = ch; a.vis.chunks_head_stride = (long)nb * cld; a.vis.chunks_ld = cld;
        }
        a.out = out; a.ldo = D;
        a.B = B; a.G = G; a.H = H; a.D = D; a.Lmax = Lmax; a.n = n; a.prefix = prefix; a.scale = 0.125f;
        a.trace = ctx->trace;
        if (split == -1) { a.stage_cap = 0; split = 0; }                       // the fused kernel without K/V pieces staged in LDS
        if (split) {
            BG_REQUIRE(skinny_fused_supported(B, 3 * D, D, true) && (!w_f16 || skinny_fused_f16_ok(3 * D, D, true)), "op_ar_attn_fused: the split form does not support B=%d D=%d", B, D);
            float* wp = ctx->arena.get<float>(skinny_packed_floats(3 * D, D));
            float* qkv = ctx->arena.get<float>((size_t)B * 3 * D);
            float* xn = ctx->arena.get<float>((size_t)B * D);
            if (w_f16) launch_pack_skinny_weight_f16(wqkv, wp, 3 * D, D, s);   // rounds to fp16 while packing
            else launch_pack_skinny_weight(wqkv, wp, 3 * D, D, s);
            SkinnyFusedArgs pq;
            const RowSrc src = a.x;
            pq.a_src = &src;
            pq.ln_w = ln_w; pq.ln_b = ln_b; pq.Wp = wp; pq.w_f16 = w_f16; pq.bias = bqkv;
            pq.C = qkv; pq.ldc = 3 * D; pq.xn_out = xn; pq.ldxn = D;
            pq.M = B; pq.N = 3 * D; pq.K = D; pq.ksplit = 1;
            launch_skinny_fused(pq, s);
            a.qkv = qkv; a.xn = xn;
            if (split > 1) {   // split = k > 1: the attention-only kernel with its key walk cut into k ranges (one sequence per workgroup)
                BG_REQUIRE(G == 1, "op_ar_attn_fused: a key split needs G = 1");
                a.ksplit = split;
                a.kws = ctx->arena.get<float>((size_t)B * H * split * 66);
            }
        }
        launch_ar_attn_fused(a, s);
    });
}

int bevgen_op_conv3x3(bevgen_ctx* ctx, const float* x, const float* w, const float* bias, const float* residual, float* y, int n, int H, int W, int Cin, int Cout,
                      int up, void* stream) {
    return guarded(ctx, [&] {
        GemmArgs g;
        const int flags = up;   // bit 0: nearest-2x upsample fused into the gather; bit 1: the LDS-DMA kernel on operand planes split here (tests / probes: the model hands it
        up &= 1;                // planes its GroupNorm wrote); bit 2 (with bit 1): the general convolution variant also where the stride-1 one (MODE_CONV3S) applies
        const int oh = up ? 2 * H : H, ow = up ? 2 * W : W;
        g.mode = MODE_CONV3;
        g.A = x; g.B = w; g.C = y; g.R = residual; g.bias_n = bias;
        g.M = n * oh * ow; g.N = Cout; g.K = 9 * Cin; g.lda = Cin; g.ldb = 9 * Cin; g.ldc = Cout; g.ldr = Cout;
        g.conv_h = oh; g.conv_w = ow; g.conv_cin = Cin; g.conv_up = up;
        if (flags & 2) {
            BG_REQUIRE(Cin % 32 == 0, "op_conv3x3: the LDS-DMA kernel needs Cin %% 32 == 0 (Cin=%d)", Cin);
            const size_t xb = (size_t)n * H * W * Cin * 4, wb = (size_t)Cout * 9 * Cin * 4;
            ctx->arena.reserve(xb + wb + 4096);
            ctx->arena.reset();
            uint16_t* ap = reinterpret_cast<uint16_t*>(ctx->arena.alloc(xb));
            uint16_t* bp = reinterpret_cast<uint16_t*>(ctx->arena.alloc(wb));
            launch_split_weight(x, ap, (long)n * H * W * Cin, (hipStream_t)stream);
            launch_split_weight(w, bp, (long)Cout * 9 * Cin, (hipStream_t)stream);
            g.A = nullptr; g.A_hi = ap; g.A_lo = ap + 32; g.B_hi = bp; g.B_lo = bp + 32;
            g.conv_general = (flags & 4) != 0;
        }
        launch_gemm(g, (hipStream_t)stream);
    });
}

int bevgen_op_groupnorm(bevgen_ctx* ctx, const float* x, const float* gamma, const float* beta, float* y, int n, int hw, int C, int swish, void* stream) {
    return guarded(ctx, [&] {
        ctx->arena.reserve(groupnorm_ws_bytes(n, hw) + (size_t)n * 64 * sizeof(float) + 1024);
        ctx->arena.reset();
        float* stats = ctx->arena.get<float>((size_t)n * 64);
        void* ws = ctx->arena.alloc(groupnorm_ws_bytes(n, hw));
        launch_groupnorm_stats(x, stats, ws, n, hw, C, 1e-6f, (hipStream_t)stream);
        launch_groupnorm_apply(x, stats, gamma, beta, y, n, hw, C, swish, (hipStream_t)stream);
    });
}

int bevgen_decode_attention_splits(int B, int H, int n) { return decode_attention_splits(B, H, n); }

int bevgen_profile_begin(bevgen_ctx* ctx) {
    return guarded(ctx, [&] { ctx->prof.begin(); });
}

int bevgen_profile_end(bevgen_ctx* ctx, double* out) {
    return guarded(ctx, [&] {
        BG_REQUIRE(out, "profile_end: null output");
        ctx->prof.end(out);
    });
}

int bevgen_ar_step_timing(bevgen_ctx* ctx, int enable) {
    return guarded(ctx, [&] { ctx->time_steps = enable != 0; });
}

int bevgen_ar_step_times(bevgen_ctx* ctx, float* h_out_ms, int cap, int* count) {
    return guarded(ctx, [&] {
        BG_REQUIRE(h_out_ms && count && cap >= 0, "ar_step_times: bad arguments");
        *count = ar_step_times(*ctx, h_out_ms, cap);
    });
}

int bevgen_set_trace_buffer(bevgen_ctx* ctx, void* d_buf) {
    return guarded(ctx, [&] { ctx->trace = reinterpret_cast<long long*>(d_buf); });
}

}  // extern "C"
