/* libbevgen_hip - C ABI of the MI355X-native BEVGen stage-2 sampling path (gfx950 / ROCm).
 *
 * The reference (alexanderswerdlow/BEVGen) is pure Python: it has NO FFI for this path; its "plugin" mechanism is
 * Hydra `_target_` instantiation of Python classes (configs/experiment/muse_stage_two_multi_view.yaml:17,26,31,41;
 * configs/model/stage_2.yaml:1,7,9,36,59).  The drop-in is therefore two layers:
 *   - Python classes with the reference's constructor/forward signatures (package bevgen_amd), and
 *   - this C ABI underneath them, which is what a maintainer binds (ctypes stub in INTEGRATION.md).
 * Each entry point below names the reference function whose arithmetic it replaces (paths under
 * multi_view_generation/, line numbers as in the surveyed tree).
 *
 * Conventions
 *   - every function returns 0 on success, a negative code on failure; bevgen_last_error(ctx) gives the message
 *   - the CALLER owns every buffer passed in; `const T* d_*` / `T* d_*` are DEVICE pointers (e.g. torch tensor
 *     data_ptr()), `h_*` are HOST pointers; the library owns weights, KV cache and workspace inside the context
 *   - `stream` is a hipStream_t (0 = default stream); calls are asynchronous on it unless stated otherwise,
 *     the library performs no hidden host synchronisation on the sampling path.  One documented exception: a Route-A context created with decode_path = AUTO
 *     and max_batch > 4 that is nevertheless called with <= 4 sequences packs the split decode layer's q/k/v operand image on that FIRST such call
 *     (one device allocation per layer, ~300 MB at BASELINE config 4 in fp32, and one stream synchronisation, once per context); every other
 *     configuration (max_batch 0 or <= 4, decode_path = SPLIT) takes it at bevgen_finalize
 *   - one context per (device, model); a context is not thread-safe, different contexts may be used from different host threads concurrently
 *   - int64 token ids, fp32 everything else (the arithmetic type is a context property: BEVGEN_PRECISION_*)
 */
#ifndef BEVGEN_HIP_H
#define BEVGEN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BEVGEN_ABI_VERSION 6   /* 6: device status word (bevgen_synchronize, bevgen_status, BEVGEN_ERR_NUMERIC, BEVGEN_STATUS_*).  5: bevgen_op_mlp_fused; a context without max_batch <= 4 packs the split decode layer lazily (see Conventions).  4: 4: BEVGEN_PROFILE_KINDS 5 -> 6 (bevgen_profile_end writes 18 doubles), bevgen_cfg.decode_chains, decode_path values 2 / 3 */

enum { BEVGEN_ROUTE_MASKGIT = 0, BEVGEN_ROUTE_AR = 1 };
/* FP32  : every product and accumulation in exact fp32 on the matrix cores (v_mfma_f32_32x32x2_f32) - bit-exact greedy tokens vs the CPU reference.
 * F16X3 : large GEMMs / convolutions as 3 f16 MFMAs per k-step on (hi, lo*2^-11) splits of the fp32 operands: ~2^-22 relative error per product,
 *         fp32 accumulation, up to 5x the fp32 MFMA rate.  Everything else (attention, norms, samplers, decode-step GEMMs) stays fp32. */
enum { BEVGEN_PRECISION_FP32 = 0, BEVGEN_PRECISION_BF16 = 1 /* reserved */, BEVGEN_PRECISION_F16X3 = 2 };
enum { BEVGEN_KV_F32 = 0, BEVGEN_KV_F16 = 1 };
enum { BEVGEN_DECODE_FUSED = 0, BEVGEN_DECODE_PER_OP = 1, BEVGEN_DECODE_SPLIT = 2, BEVGEN_DECODE_AUTO = 3 };
enum { BEVGEN_W_F32 = 0, BEVGEN_W_F16 = 1 };
enum { BEVGEN_DTYPE_F32 = 0, BEVGEN_DTYPE_I64 = 1, BEVGEN_DTYPE_U8 = 2, BEVGEN_DTYPE_F64 = 3 };

enum {
    BEVGEN_OK = 0,
    BEVGEN_ERR_INVALID = -1,   /* bad argument / missing tensor / unsupported size */
    BEVGEN_ERR_HIP = -2,       /* a HIP runtime call failed */
    BEVGEN_ERR_STATE = -3,     /* call order violated (e.g. generate before finalize) */
    BEVGEN_ERR_INTERNAL = -4,
    BEVGEN_ERR_NUMERIC = -5    /* a kernel flagged a non-finite logit / pixel or an operand outside the f16 range of the chosen precision (device status word, below):
                                  the results of the call that raised it are invalid */
};

/* Device status word.  The reference asserts on the HOST, every decode step, that the transformer's inputs and logits are finite
 * (transformer/mingpt_sparse.py:383,388; stage2/cond_transformer_multi_view.py:202).  This library never synchronises on the sampling path; instead the kernels
 * that touch the values anyway OR a bit into one host-visible word per context (mapped host memory) and the host reads it
 *   - in bevgen_synchronize (after waiting for the stream: the call a caller makes before it trusts / copies the results),
 *   - at the entry of EVERY later entry point on the context (no synchronisation: an error raised by an earlier asynchronous call is reported by the next call at the latest),
 *   - in bevgen_finalize (weights outside the f16 range of the chosen precision) and, as a message on stderr, in bevgen_destroy.
 * A raised word is returned as BEVGEN_ERR_NUMERIC (bits 4, 8, 16) or BEVGEN_ERR_INTERNAL (bits 1, 2) and then cleared.
 * Policy for precision = F16X3 / fp16 storage modes: an operand whose f16 image would be inf / NaN (|v| >= 65520) is an ERROR, not rescaled - the split keeps fp32-class
 * mantissas but f16's exponent range, where the reference's bf16 / fp32 arithmetic has 8 exponent bits; such a checkpoint must run with BEVGEN_PRECISION_FP32. */
enum {
    BEVGEN_STATUS_MLP_BARRIER = 1,        /* fused MLP launch of the decode step: an XCD-local barrier timed out (the GPU was shared); the context falls back to two launches */
    BEVGEN_STATUS_MLP_PLACEMENT = 2,      /* ...: a workgroup was not placed on the XCD its index implies */
    BEVGEN_STATUS_NONFINITE_LOGITS = 4,   /* a sampler read a NaN / inf logit or critic score */
    BEVGEN_STATUS_F16_RANGE = 8,          /* a value written as an f16 operand (hi / lo planes, fp16 KV cache, fp16 decode activations, fp16 weights) was NaN or |v| >= 65520 */
    BEVGEN_STATUS_NONFINITE_PIXELS = 16   /* the VQGAN decoder produced a NaN / inf pixel */
};

/* Sizes of the stage-2 transformer (mirror of GPTConfig, modules/transformer/mingpt_sparse.py:26-102) and of the
 * stage-1 VQGAN decoder (ddconfig, configs/model/stage_2.yaml:45-55).  Zero-initialise, then fill. */
typedef struct bevgen_cfg {
    int32_t abi_version;     /* = BEVGEN_ABI_VERSION */
    int32_t route;           /* BEVGEN_ROUTE_* */
    int32_t precision;       /* BEVGEN_PRECISION_* (arithmetic of projections / KV-cache storage) */
    int32_t num_layers, num_heads, dim, vocab_size, cond_vocab_size;
    int32_t num_cams, cam_latent_h, cam_latent_w;      /* T = h*w tokens per camera, N = C*T */
    int32_t num_cond_tokens;                           /* K (BEV latent cells) */
    int32_t seq_len;                                   /* L = gpt_block_size */
    int32_t sparse_block_size;                         /* blk; L % blk == 0 */
    int32_t image_embed, bev_embed, camera_bias;       /* feature flags */
    int32_t ff_inner;                                  /* Route M: int(dim*4*2/3); 0 for Route A */
    int32_t max_batch;                                 /* upper bound on scenes (Route M) / sequences (Route A) per call */
    /* stage-1 decoder; vq_ch == 0 -> no VQGAN in this context */
    int32_t vq_ch, vq_num_res_blocks, vq_z_channels, vq_embed_dim, vq_n_embed, vq_resolution, vq_out_ch;
    int32_t vq_num_levels;
    int32_t vq_ch_mult[8];
    int32_t vq_attn_resolution;                        /* spatial size at which AttnBlocks are inserted (16) */
    int32_t vq_in_channels;                            /* encoder input channels (3 images / 7 Argoverse BEV classes); 0 = decoder only */
    int32_t kv_cache_dtype;                            /* Route A KV-cache storage: BEVGEN_KV_F32 (default, bit-exact tokens) or BEVGEN_KV_F16 (fp16 storage,
                                                          fp32 accumulate: half the decode-attention HBM traffic; tokens no longer guaranteed identical) */
    int32_t decode_path;                               /* Route A decode step: BEVGEN_DECODE_FUSED (default: three launches per layer, decode_fused.hip) or
                                                          BEVGEN_DECODE_PER_OP (one kernel per operator: the round-1 path, kept as the A/B reference) or
                                                          BEVGEN_DECODE_SPLIT (four launches per layer: LayerNorm + QKV projection of the whole batch as one MFMA kernel that
                                                          reads the weight once, then the decode-attention kernel proper = the K/V stream and nothing else) or
                                                          BEVGEN_DECODE_AUTO (what the Python host asks for: SPLIT, with the key walk of every (sequence, head) cut into up to
                                                          four ranges, for up to four sequences / layout groups per call - the interactive single-scene caller, where 16-64
                                                          workgroups per layer cannot pull weights and K/V fast enough: 0.92 vs 1.12 ms/step at B = 1 - else FUSED; with
                                                          decode_weight_dtype = BEVGEN_W_F16 FUSED at every batch size: its single MLP launch makes it the faster form, 0.71 vs 0.83) */
    int32_t decode_weight_dtype;                       /* Route A projection weights (q/k/v, MLP, head): BEVGEN_W_F32 (default) or BEVGEN_W_F16: bevgen_finalize rounds them to
                                                          fp16-representable values (prefill and decode then use the same model: the reference's Route A runs fp16,
                                                          sparse_self_attention.py:127) and the decode step streams the 2-byte copies: half the weight traffic */
    int32_t weight_dtype;                              /* every matrix that feeds a split-precision GEMM / convolution (needs BEVGEN_PRECISION_F16X3): BEVGEN_W_F32
                                                          (default: hi + lo f16 planes, three MFMAs per product, fp32-class results on the fp32 weights) or BEVGEN_W_F16:
                                                          bevgen_finalize rounds those matrices to f16 once (the reference's own GPU runs cast them to bf16 / fp16 on
                                                          every call: muse bf16 autocast, sparse_self_attention.py:127) - activations keep their hi + lo planes, a
                                                          product is two MFMAs, and the results are fp32-class results OF THE ROUNDED MODEL.  Route A: together with
                                                          decode_weight_dtype = BEVGEN_W_F16 only */
    int32_t decode_chains;                             /* Route A fused decode step: the batch is cut into this many independent sequence groups ("chains") whose
                                                          layer kernels are enqueued on separate HIP streams (one fork / join per step inside the captured graph), so
                                                          that one chain's weight-streaming projections run under another chain's K/V stream instead of every short
                                                          dependent kernel paying its ramp alone.  0 = the library's choice for the batch, 1 = a single chain */
    int32_t reserved[10];
} bevgen_cfg;

typedef struct bevgen_ctx bevgen_ctx;

/* ---------------------------------------------------------------------------------------------------------------
 * lifecycle                                                                                                       */
int bevgen_create(const bevgen_cfg* cfg, int device, bevgen_ctx** out);
void bevgen_destroy(bevgen_ctx* ctx);
const char* bevgen_last_error(const bevgen_ctx* ctx);   /* ctx may be NULL: error of the last failed bevgen_create */
int bevgen_abi_version(void);
/* Wait for `stream` (and the library's own decode stream) and report what the kernels of the calls enqueued so far flagged: 0, or BEVGEN_ERR_NUMERIC / BEVGEN_ERR_INTERNAL
 * with the details in bevgen_last_error.  This is where the reference's per-step `assert isfinite` lands (see "Device status word" above). */
int bevgen_synchronize(bevgen_ctx* ctx, void* stream);
/* The raw status word (BEVGEN_STATUS_* bits) as the host sees it right now: no synchronisation, nothing cleared. */
int bevgen_status(bevgen_ctx* ctx, unsigned* word);

/* Upload one tensor (synchronous H2D copy).  `name` is the reference state_dict key
 * (utils/general.py:119-160 loader semantics: names listed in bevgen_amd/weights.py), with the prefixes
 *   "transformer." / "token_critic."  Route M MaskGit keys   (stage2/muse_maskgit_pytorch.py:204-261, 388-392)
 *   ""                                Route A GPT keys        (transformer/mingpt_sparse.py:267-308)
 *   "first_stage_model."              stage-1 VQModel keys    (stage1/vqgan.py:31-80)
 * or one of the static tables (built on the host exactly as in the reference, bevgen_amd/tables.py):
 *   "table.forward_shuffle_idx" i64[N]   CustomPermuter          transformer/permuter.py:33-88
 *   "table.attention_mask"      f32[L,L] allowed_pattern[0]      transformer/mask_generator.py:202-206
 *   "table.layout"              i64[H,L/blk,L/blk]               transformer/mask_generator.py:217-228
 *   "table.prob_matrix"         f32[L,L] camera-bias prior       transformer/mask_generator.py:172-190
 *   "table.image_plane"         f32[3,T] pixel plane             transformer/mingpt_sparse.py:288-294
 * Unknown names are stored and ignored (strict=False). */
int bevgen_load_tensor(bevgen_ctx* ctx, const char* name, const void* h_data, int dtype, int ndim, const int64_t* shape);

/* Convenience wrapper: the four tables at once (any pointer may be NULL to skip it). */
int bevgen_set_tables(bevgen_ctx* ctx, const int64_t* h_forward_shuffle_idx, const float* h_attention_mask,
                      const int64_t* h_layout, const float* h_prob_matrix, const float* h_image_plane);

/* Validate that every tensor the route needs is present and build the derived device data (fused QKV weights,
 * attention-bias matrices = tril-scatter(camera_bias_emb) + prob_matrix as in mingpt_sparse.py:375-380 /
 * muse_maskgit_pytorch.py:343-348, visibility masks, re-laid-out conv kernels).  Must precede any compute call. */
int bevgen_finalize(bevgen_ctx* ctx);

/* ---------------------------------------------------------------------------------------------------------------
 * Route M - MaskGit                                                                                               */

/* TransformerMultiView.forward in eval mode (stage2/muse_maskgit_pytorch.py:283-371):
 *   d_ids [B*C, T] (mask id = vocab_size), d_cond_ids [B, K], d_I_inv [B,C,3,3], d_E_inv [B,C,4,4]
 *   -> d_logits [B*C, T, V] and/or d_embed [B*C, T, D] (either may be NULL). */
int bevgen_muse_forward(bevgen_ctx* ctx, const int64_t* d_ids, const int64_t* d_cond_ids, const float* d_I_inv, const float* d_E_inv,
                        int B, float* d_logits, float* d_embed, void* stream);

/* MaskGit.generate with the self token critic (stage2/muse_maskgit_pytorch.py:511-627).
 *   h_mask_schedule[timesteps]  tokens re-masked at each iteration, max(int(cos(pi/2 t) T), 1) (:566-567) - host-computed
 *   topk_k                      ceil((1 - topk_filter_thres) * V) (:453-454)
 *   d_gumbel_u [timesteps, B*C, T, V], d_critic_u [timesteps, B*C, T]  explicit U[0,1) noise replacing :430-431/:446-448,
 *                               NULL = deterministic setting (gumbel noise 0, critic uniform 0.5)
 *   noise_seed                  used when the explicit tensors are NULL: != 0 -> the samplers draw their uniforms in registers (Philox4x32-10 keyed by
 *                               (noise_seed, iteration, element): stream 0 = gumbel [rows*V], stream 1 = critic [rows]; bevgen_op_philox_uniform writes the
 *                               same numbers out) - stochastic sampling without 1.8 GB of noise tensors; 0 -> the deterministic setting
 *   d_init_ids [B*C, T] or NULL (partial decoding, :543-544, 573-574)
 *   -> d_out_ids [B*C, T]
 * The classifier-free-guidance "null" forwards of the reference (:272-276, 394-396) are bit-identical to the conditional
 * ones in eval mode and are not executed; the critic forward after the last iteration (result unused) is skipped. */
int bevgen_maskgit_generate(bevgen_ctx* ctx, const int64_t* d_cond_ids, const float* d_I_inv, const float* d_E_inv, int B,
                            int timesteps, const int32_t* h_mask_schedule, float temperature, int topk_k, float critic_noise_scale,
                            const float* d_gumbel_u, const float* d_critic_u, const int64_t* d_init_ids, int64_t* d_out_ids,
                            unsigned long long noise_seed, void* stream);

/* The remaining branches of MaskGit.generate (stage2/muse_maskgit_pytorch.py:511-627) and BASELINE config 5's sample sharing:
 *   score_mode 0  the token critic (above);
 *              1  force_not_use_token_critic = True (:611-619): scores = 1 - softmax(logits)[pred], -1e5 at positions that were not masked in the iteration;
 *                 no critic forward is run (18 transformer forwards per call instead of 35); d_critic_u / critic_noise_scale are not used;
 *              2  ... with can_remask_prev_masked = True (:620-622): the same scores at EVERY position (the prediction is drawn everywhere)
 *   samples_per_layout S > 1: consecutive groups of S scenes share their BEV layout and cameras (all B = layouts * S rows of d_cond_ids / matrices are
 *                 passed; the flag enables the reuse): the condition embedding and the cross-attention K / V of every layer are built once per
 *                 LAYOUT and read by the S samples of the group. */
int bevgen_maskgit_generate_ex(bevgen_ctx* ctx, const int64_t* d_cond_ids, const float* d_I_inv, const float* d_E_inv, int B,
                               int timesteps, const int32_t* h_mask_schedule, float temperature, int topk_k, float critic_noise_scale,
                               const float* d_gumbel_u, const float* d_critic_u, const int64_t* d_init_ids, int64_t* d_out_ids,
                               unsigned long long noise_seed, int score_mode, int samples_per_layout, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Route A - autoregressive sparse-causal transformer with camera bias (prefill + KV-cache decode)                 */

/* Operator seam: SparseSelfAttention.forward(query, key, value, attn_mask, add_mask)
 * (transformer/sparse_self_attention.py:103-177): q,k,v,out [B,H,L,64] fp32, layout i64 [H,L/blk,L/blk],
 * attn_mask f32 [L,L] ('mul' mode: 0 = masked), add_mask f32 [L,L] or NULL.  Dense restatement of the DeepSpeed
 * sdd/softmax/dsd pipeline: out = softmax(dh^-0.5 (q k^T + add_mask) + M) v.  Stateless (ctx only carries errors/workspace). */
int bevgen_sparse_self_attention(bevgen_ctx* ctx, const float* d_q, const float* d_k, const float* d_v, const int64_t* d_layout,
                                 const float* d_attn_mask, const float* d_add_mask, int B, int H, int L, int block, float* d_out, void* stream);

/* GPT.forward over the K condition rows (transformer/mingpt_sparse.py:319-391 restricted to rows 0..K-1): fills the KV cache
 * of all layers for B sequences and leaves the hidden state of row K-1 ready for bevgen_ar_logits. */
int bevgen_ar_prefill(bevgen_ctx* ctx, const int64_t* d_cond_ids, const float* d_I_inv, const float* d_E_inv, int B, void* stream);

/* ln_f + head on the newest row (mingpt_sparse.py:386-387): d_logits [B, V] = logits of decode step `n_decoded`. */
int bevgen_ar_logits(bevgen_ctx* ctx, float* d_logits, void* stream);

/* Feed the token chosen at the current step (stage2/cond_transformer_multi_view.py:219) and advance one position:
 * embedding of the token at its (camera, latent cell) + one pass over all layers against the KV cache. */
int bevgen_ar_decode_step(bevgen_ctx* ctx, const int64_t* d_token, void* stream);

/* Net2NetTransformer.sample (stage2/cond_transformer_multi_view.py:154-227) with prefill + KV cache:
 *   top_k <= 0 disables the filter; greedy != 0 -> arg-max (sample=False), else inverse-CDF draw with d_noise_u [steps, B];
 *   samples_per_layout: consecutive groups of sequences share cond ids/cameras (B = layouts * samples_per_layout rows are
 *   passed explicitly; the flag only enables prefix reuse);  -> d_out_ids [B, C, T] (camera-major, like the reference's x) */
int bevgen_ar_sample(bevgen_ctx* ctx, const int64_t* d_cond_ids, const float* d_I_inv, const float* d_E_inv, int B, int steps,
                     int top_k, float temperature, int greedy, const float* d_noise_u, int samples_per_layout,
                     int64_t* d_out_ids, float* d_step_logits /* [steps,B,V] or NULL */, void* stream);

/* Same with partial decoding (cond_transformer_multi_view.py:161-165, 181-182: the tokens of the cameras in `partial_decoding_idx` are taken from the
 * encoded ground-truth images and their positions are skipped by the sampling loop): d_forced_ids [steps, B] int64 in DECODE order, entry >= 0 = the
 * token to emit at that step (it is still pushed through the stack so that later positions attend to it), < 0 = draw the token.  NULL = bevgen_ar_sample. */
int bevgen_ar_sample_forced(bevgen_ctx* ctx, const int64_t* d_cond_ids, const float* d_I_inv, const float* d_E_inv, int B, int steps,
                            int top_k, float temperature, int greedy, const float* d_noise_u, int samples_per_layout,
                            const int64_t* d_forced_ids, int64_t* d_out_ids, float* d_step_logits, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * stage-1 VQGAN decode                                                                                            */

/* decode_to_img (stage2/cond_transformer_multi_view_muse.py:157-164): get_codebook_entry (stage1/quantize.py:314-329) ->
 * post_quant_conv + Decoder (stage1/vqgan.py:118-121, stage1/model.py:506-537) [-> util.denormalize_tensor, bev_utils/util.py:97-118].
 * The decoder is fully convolutional: the latent grid is lat_h x lat_w = cam_latent_res (16 x 16 for 256 x 256 images, 14 x 25 for nuScenes 224 x 400;
 * 0, 0 = the square grid of ddconfig.resolution), the image (lat_h << (levels-1)) x (lat_w << (levels-1)).
 *   out_mode BEVGEN_VQ_OUT_RAW      d_out fp32 [n, out_ch, H, W], the decoder output
 *            BEVGEN_VQ_OUT_DENORM   d_out fp32 [n, 3, H, W] in [0,1]  (x*std + mean, clamped)
 *            BEVGEN_VQ_OUT_U8       d_out uint8 [n, 3, H, W] = round(255 * denormalised): the storage / wire format of the generated images
 *   d_ids [n, lat_h*lat_w]. */
enum { BEVGEN_VQ_OUT_RAW = 0, BEVGEN_VQ_OUT_DENORM = 1, BEVGEN_VQ_OUT_U8 = 2 };
int bevgen_vq_decode(bevgen_ctx* ctx, const int64_t* d_ids, int n, int lat_h, int lat_w, int out_mode, void* d_out, void* stream);

/* VQModel.encode (stage1/vqgan.py:84-116 with geometric_embedding=False): Encoder (stage1/model.py:405-433) -> quant_conv ->
 * VectorQuantizer2.forward arg-min (stage1/quantize.py:271-312).  d_x [n, in_channels, H, W] fp32 (NCHW, as get_input produces it,
 * muse_lm:166-179; H, W multiples of 2^(levels-1); 0, 0 = ddconfig.resolution squared) -> d_ids [n, (H >> (levels-1)) * (W >> (levels-1))] int64.
 * This is encode_to_c (BEV segmentation -> condition tokens) and encode_to_z (muse_lm:142-155). */
int bevgen_vq_encode(bevgen_ctx* ctx, const float* d_x, int n, int H, int W, int64_t* d_ids, void* stream);

/* VQModel.decode(quant) (stage1/vqgan.py:118-121) for already looked-up latents: d_zq [n, embed_dim, lat_h, lat_w] (NCHW, like the reference). */
int bevgen_vq_decode_latents(bevgen_ctx* ctx, const float* d_zq, int n, int lat_h, int lat_w, int out_mode, void* d_out, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Operator-level entry points (parity tests and roofline measurements call the kernels through these)             */
int bevgen_op_gemm(bevgen_ctx* ctx, const float* d_a, const float* d_w, const float* d_bias, const float* d_residual, float* d_c,
                   int M, int N, int K, int act_gelu, int skinny, void* stream);            /* C = A W^T (+bias)(gelu)(+res); skinny: 0 tiled fp32, 1 M <= 64, 2-4 split-precision
                                                                                               kernels (tests / probes), 5 = 3 with the k range split over three slices, 6 = 3 in its stream-K form */
/* Decode-step projection (decode_fused.hip): C = act(LayerNorm?(A) W^T + bias) for M <= 64 rows; d_ln_w NULL = no LayerNorm; ksplit > 1 (no LayerNorm, no bias / act):
 * d_c receives the split-K partial sums [ksplit, M, N] that the consumer adds; ksplit 0 = the library's choice for (N, K), returned through *ksplit_out if non-NULL;
 * ksplit -1 (LayerNorm with beta and bias): one K slice, the LayerNorm folded into the product the way the decode step launches ln2 + MLP-up. */
int bevgen_op_ln_gemm(bevgen_ctx* ctx, const float* d_a, const float* d_ln_w, const float* d_ln_b, float eps, const float* d_w, const float* d_bias, float* d_c,
                      int M, int N, int K, int act_gelu, int ksplit, int* ksplit_out, void* stream);
/* Both MLP projections of a Route-A decode layer in one launch (decode_fused.hip ar_mlp_fused_kernel; Block.forward's mlp(ln2(x)), transformer/mingpt_sparse.py:232-253):
 * d_out [M, D] = Linear2(GELU(Linear1(LayerNorm(d_x)))) WITHOUT the residual, M <= 64, D = 1024; w_f16: the matrices are rounded to fp16 first (decode_weights = f16).
 * Returns BEVGEN_ERR_INVALID where the launch is not supported (shape, or a device with fewer CUs than its 4 D / 16 workgroups). */
int bevgen_op_mlp_fused(bevgen_ctx* ctx, const float* d_x, const float* d_ln_w, const float* d_ln_b, float eps, const float* d_w1 /*[4D, D]*/, const float* d_b1,
                        const float* d_w2 /*[D, 4D]*/, const float* d_b2, int w_f16, float* d_out, int M, int D, void* stream);
/* d_out[i] = the uniform the MaskGit samplers draw for element i of noise stream `stream_id` (0 gumbel, 1 critic) at iteration `iter` under `seed`. */
int bevgen_op_philox_uniform(bevgen_ctx* ctx, unsigned long long seed, unsigned iter, unsigned stream_id, int V /* vocabulary size: row layout of stream 0 */, long n,
                             float* d_out, void* stream);
int bevgen_op_layernorm(bevgen_ctx* ctx, const float* d_x, const float* d_gamma, const float* d_beta, float* d_y, int rows, int D, float eps, void* stream);
int bevgen_op_geglu_layernorm(bevgen_ctx* ctx, const float* d_h, const float* d_gamma, float* d_y, int rows, int F, int ldy, void* stream);
int bevgen_op_attention(bevgen_ctx* ctx, const float* d_q, const float* d_k, const float* d_v, const float* d_bias, int ldbias,
                        int B, int H, int Nq, int Nk_pad, float scale, float* d_out, void* stream);   /* q [B,H,Nq,64], k/v [B,H,Nk_pad,64] */
/* The same with the key tiles of the split-precision kernel cut into `key_splits` ranges (1..8; needs Nk_pad / 32 >= key_splits) merged by a combine kernel - the form
 * the Route M model path picks for its self-attention at one scene per call (muse.cpp pick_attn_ksplit); fp32-precision contexts ignore key_splits. */
int bevgen_op_attention_ex(bevgen_ctx* ctx, const float* d_q, const float* d_k, const float* d_v, const float* d_bias, int ldbias,
                           int B, int H, int Nq, int Nk_pad, float scale, int key_splits, float* d_out, void* stream);
int bevgen_op_decode_attention(bevgen_ctx* ctx, const float* d_q, const void* d_kcache, const void* d_vcache, int kv_dtype,
                               const float* d_bias, int ldbias, const uint8_t* d_keep, int ldkeep, long keep_head_stride,
                               int B, int H, int n, int Lmax, float scale, float* d_out, void* stream);
/* Attention half of one Route A decode layer as the fused path runs it (decode_fused.hip ar_attn_fused_kernel; Block.forward mingpt_sparse.py:240-253 +
 * SparseSelfAttention.forward sparse_self_attention.py:150-176) for ONE new row per sequence:
 *   row = d_x[b] (+ d_rbias + sum of the ns split-K partials d_partial [ns, B, D]) -> xn = ln1(row) -> q | k | v = Wqkv xn + bqkv (fused [3D, D]) ->
 *   k, v written to cache row n-1 -> softmax(dh^-0.5 (q k^T + d_bias[n-1, :]) + mask) v over keys 0..n-1 -> d_out[b] = xn + attention.
 * Visibility as the reference defines it: d_attn_mask [L, L] fp32 (0 = hidden) AND d_layout int64 [H, L/block, L/block] (0 = block absent, never read);
 * either may be NULL.  G > 1: consecutive groups of G sequences share their first `prefix` keys, read from the group's first cache slot.
 * kv_dtype 0 fp32 / 1 fp16 cache [B, H, Lmax, 64]; w_f16 != 0: the projection weights are rounded to fp16 and streamed as 2-byte values.
 * split = 0: the fused kernel as the sampling path launches it (G = 1: leading K/V steps of the key walk staged in LDS by LDS-DMA); split = -1: without staged pieces.
 * split > 0: the BEVGEN_DECODE_SPLIT form of the same computation (LayerNorm + QKV projection kernel, then the attention-only kernel); split = k > 1: with the
 * key walk of every (sequence, head) cut into k ranges on k workgroups and merged by the combine kernel (what one- and two-sequence calls run; G = 1). */
int bevgen_op_ar_attn_fused(bevgen_ctx* ctx, const float* d_x, const float* d_partial, int ns, const float* d_rbias, const float* d_ln_w, const float* d_ln_b,
                            const float* d_wqkv, const float* d_bqkv, int w_f16, void* d_kcache, void* d_vcache, int kv_dtype, const float* d_bias, int ldbias,
                            const float* d_attn_mask, const int64_t* d_layout, int block, int B, int G, int H, int n, int Lmax, int prefix, int split, float* d_out, void* stream);
/* upsample2x: bit 0 = nearest-2x upsample of the input fused into the gather; bit 1 = run the LDS-DMA kernel on (hi, lo) operand planes split inside the call (Cin % 32 == 0;
 * the decoder reaches that kernel with planes its GroupNorm wrote); bit 2 (with bit 1) = its general variant also where the stride-1 variant applies. */
int bevgen_op_conv3x3(bevgen_ctx* ctx, const float* d_x_nhwc, const float* d_w_ohwi, const float* d_bias, const float* d_residual,
                      float* d_y_nhwc, int n, int H, int W, int Cin, int Cout, int upsample2x, void* stream);
int bevgen_op_groupnorm(bevgen_ctx* ctx, const float* d_x_nhwc, const float* d_gamma, const float* d_beta, float* d_y, int n, int hw, int C,
                        int swish, void* stream);

/* Per-kernel timing support for bench.py's roofline leg: number of workgroup splits the decode-attention kernel uses. */
int bevgen_decode_attention_splits(int B, int H, int n);

/* HIP-event timing of the hot kernels on their launch stream.  Between begin and end every launch of
 *   0 gemm (fp32 MFMA)   1 conv3x3 (implicit GEMM)   2 flash attention   3 decode attention   4 skinny GEMM
 *   5 small-problem launches of the split-precision GEMM (128- / 64-row blocks, the short last part of a row-split launch; kind 0 is then one kernel)
 * is bracketed by an event pair; end synchronises the device and writes out[kind*3 + {0,1,2}] =
 * {launches, total milliseconds, total algorithmic work (FLOP for 0-2 and 5, bytes for 3-4)} for the 6 kinds (18 doubles). */
#define BEVGEN_PROFILE_KINDS 6
int bevgen_profile_begin(bevgen_ctx* ctx);
int bevgen_profile_end(bevgen_ctx* ctx, double* out);

/* Per-step latency of bevgen_ar_sample: with timing enabled the library records one HIP event after every replay of the captured decode step;
 * bevgen_ar_step_times writes the durations (ms) of the most recent call's steps to the HOST array h_out_ms[cap] and their number to *count (synchronises). */
int bevgen_ar_step_timing(bevgen_ctx* ctx, int enable);
int bevgen_ar_step_times(bevgen_ctx* ctx, float* h_out_ms, int cap, int* count);

/* Diagnostics: phase timestamps of the fused Route A decode kernels.  d_buf = device buffer of 3 * 4096 * 8 int64 (kinds: 0 ln1+qkv+attention,
 * 1 ln2+MLP-up, 2 MLP-down; [workgroup][8] 100 MHz device timestamps at the phase boundaries of the most recent launch), NULL = off. */
int bevgen_set_trace_buffer(bevgen_ctx* ctx, void* d_buf);

#ifdef __cplusplus
}
#endif
#endif /* BEVGEN_HIP_H */
