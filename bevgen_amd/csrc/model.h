// Context and model-level orchestration of libbevgen_hip (internal).
#pragma once
#include <atomic>
#include <map>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/bevgen_hip.h"
#include "common.h"
#include "kernels.h"
#include "profiler.h"

namespace bevgen {

struct DevTensor {
    void* ptr = nullptr;
    int dtype = 0;
    std::vector<int64_t> shape;
    size_t bytes = 0;
    long numel() const { long n = 1; for (auto d : shape) n *= d; return n; }
    const float* f() const { return reinterpret_cast<const float*>(ptr); }
};

// Bump allocator over one device slab: workspace of a call is carved from it, nothing is allocated on the sampling path.
struct Arena {
    char* base = nullptr;
    size_t cap = 0, off = 0, high = 0;
    void reserve(size_t bytes);
    void reset() { off = 0; }
    void* alloc(size_t bytes);
    template <class T> T* get(size_t count) { return reinterpret_cast<T*>(alloc(count * sizeof(T))); }
    void release();
};

struct MuseLayer {
    // self (0) and cross (1) attention, feed-forward
    const float *norm_g[2], *to_q[2], *to_kv[2], *to_out[2], *q_scale[2], *k_scale[2], *null_kv[2];
    const float *ff_g0, *ff_w1, *ff_g3;
    float* ff_w4_padded;  // [D, Fpad], owned
    void* null_self = nullptr;      // split-precision mode: prepared null key / value of the self-attention ([k_hi|k_lo|v_hi|v_lo][H][64] halves, owned)
    float* ff_w1_geglu = nullptr;   // split-precision mode: [2 Fpad, D] rows ordered for the fused GEGLU epilogue (owned)
    float* to_qkv_self = nullptr;   // split-precision mode: to_q | to_kv of the self-attention as one [3 H 64, D] matrix (owned): one projection with the EPI_MUSE_QKV epilogue
    // LayerNorm folded across the GEMMs around it (split-precision mode, fp32 weights; GemmArgs::ln_*): W o gamma of every consumer projection (owned; registered as split
    // weights) and cs[n] = sum_k gamma_k W[n][k].  w4 = the feed-forward down-projection behind the inner LayerNorm (gamma = .2.3.gamma, zero beyond F); qkv / q_self /
    // kv_self = the self-attention projections behind norm.gamma of module 0; q_cross behind module 1's; w1 = the GEGLU up-projection behind .2.0.gamma
    float *fold_w4 = nullptr, *fold_w4_cs = nullptr;
    float *fold_qkv = nullptr, *fold_qkv_cs = nullptr, *fold_q_self = nullptr, *fold_q_self_cs = nullptr, *fold_kv_self = nullptr, *fold_kv_self_cs = nullptr;
    float *fold_q_cross = nullptr, *fold_q_cross_cs = nullptr, *fold_w1 = nullptr, *fold_w1_cs = nullptr;
};

struct ArLayer {
    const float *ln1_w, *ln1_b, *ln2_w, *ln2_b;
    float *wqkv, *bqkv;  // fused [3D, D], [3D], owned
    const float *mlp0_w, *mlp0_b, *mlp2_w, *mlp2_b;
    float *mlp0_wp = nullptr, *mlp2_wp = nullptr;   // decode-step operand images of the MLP weights (owned; fused decode path; fp32 or fp16 packed)
    void* wqkv_h = nullptr;                          // decode_weights = f16: fp16 copy of the fused QKV weight [3D, D]
    float *mlp0_cs = nullptr, *mlp0_ds = nullptr;    // the same for ln2 in front of the MLP up-projection (skinny_fused_kernel<.., FD>)
    float *ln1_cs = nullptr, *ln1_ds = nullptr;      // fused decode kernel: ln1 folded into the projection - W gamma and W beta + b of the fused QKV matrix (launch_ar_ln_fold)
    float* wqkv_wp = nullptr;                        // decode_path = split: operand image of the fused QKV weight (packed like the MLP images)
};

struct ConvW { float* w = nullptr; const float* b = nullptr; int cin = 0, cout = 0, k = 0; };  // w re-laid [Cout][kh][kw][Cin] (owned)
struct ResBlockW { const float *n1w, *n1b, *n2w, *n2b; ConvW c1, c2, nin; bool has_nin = false; int cin = 0, cout = 0; };
struct AttnBlockW { const float *nw, *nb; ConvW q, k, v, proj; int c = 0; };
struct UpLevelW { std::vector<ResBlockW> blocks; std::vector<AttnBlockW> attns; bool has_up = false; ConvW up; };
struct DownLevelW { std::vector<ResBlockW> blocks; std::vector<AttnBlockW> attns; bool has_down = false; ConvW down; };

struct Ctx {
    bevgen_cfg cfg{};
    int device = 0;
    std::string last_error;
    bool finalized = false;
    std::unordered_map<std::string, DevTensor> params;
    std::vector<void*> owned;  // derived device buffers freed at destroy
    Arena arena;               // per-call workspace
    Arena persist;             // per-batch state that must survive between C calls (Route A KV cache etc.)

    // ---- derived sizes
    int T = 0, N = 0, K = 0, L = 0, D = 0, H = 0, V = 0, F = 0, Fpad = 0;
    // ---- shared tables (device)
    const float* image_plane = nullptr;  // [3,T]
    const int64_t* fwd_idx = nullptr;    // [N]
    std::vector<int64_t> h_fwd_idx;
    float* attn_bias = nullptr;          // [L,L] = tril_scatter(camera_bias_emb) + prob_matrix (0 if !camera_bias)

    // ---- Route M
    std::vector<MuseLayer> muse;
    float *bias_self = nullptr, *bias_cross = nullptr;  // [N, ldS] / [N, ldC] with the null-key column and masks baked in
    float *bias_self_pk = nullptr, *bias_cross_pk = nullptr;  // the same two matrices as packed images for attention_split (launch_pack_attn_bias)
    int ldS = 0, ldC = 0, NkS_pad = 0, NkC_pad = 0;

    // ---- Route A
    std::vector<ArLayer> ar;
    float* head_wp = nullptr;      // packed head.weight (fused decode path)
    // visibility = element mask x block layout (kernels.h SparseVis): one [L, L] byte plane + per layer (or shared) block layouts and key-chunk lists
    uint8_t* allowed = nullptr;    // [L, L]
    uint8_t* lay = nullptr;        // [keep_layers, Hk, nb, nb]
    uint16_t* chunks = nullptr;    // [keep_layers, Hk, nb, chunks_ld] or null when no layer can skip anything (every block at or below the diagonal present)
    int chunks_ld = 0;
    bool lay_hides = true;         // some layout hides an element the mask allows (false at density 1.0: the kernels are then given no layout at all)
    int keep_heads = 1, keep_layers = 1;   // layouts per layer (1 = shared by all heads) / layers with their own layouts (1 = shared)
    float* prefill_bias = nullptr; // [Hk, K, Kpad] when the layers share a layout; else built per layer in the prefill's workspace
    int Kpad = 0;
    SparseVis vis_of_layer(int i) const {
        SparseVis v;
        const int nb = L / cfg.sparse_block_size;
        const size_t li = keep_layers > 1 ? (size_t)i : 0;
        v.allowed = allowed; v.ldallowed = L; v.allowed_head_stride = 0;
        if (lay_hides) { v.lay = lay + li * keep_heads * nb * nb; v.lay_head_stride = keep_heads > 1 ? (long)nb * nb : 0; v.nb = nb; v.blk = cfg.sparse_block_size; }
        if (chunks) { v.chunks = chunks + li * keep_heads * nb * chunks_ld; v.chunks_head_stride = keep_heads > 1 ? (long)nb * chunks_ld : 0; v.chunks_ld = chunks_ld; }
        return v;
    }
    // per-batch decode state (lives in `persist`)
    struct ArState {
        int B = 0, step = 0;        // step = number of image tokens already fed
        int G = 1;                  // sequences per layout group sharing the condition prefix rows of the group's first cache slot (fused decode path)
        float *img_embed = nullptr, *c_embed = nullptr;  // [B,C,T,D], [B,C,D]
        void *kcache = nullptr, *vcache = nullptr;       // [layers][B,H,L,64]
        float* hidden = nullptr;                         // [B,D] newest row after the last layer
        int* d_step = nullptr;                           // device copy of `step`
        bool pick_embeds = false;                        // ar_sample: the pick launch also stores the token and writes the new row's embedding (ArPickTail)
    } ars;

    // ---- VQGAN decoder
    bool has_vq = false;
    const float* codebook = nullptr;
    ConvW post_quant, conv_in, conv_out;
    ResBlockW mid1, mid2;
    AttnBlockW mid_attn;
    std::vector<UpLevelW> up;  // index = level (0 = full resolution)
    const float *norm_out_w = nullptr, *norm_out_b = nullptr;
    float *denorm_mean = nullptr, *denorm_std = nullptr;
    // ---- VQGAN encoder + quantizer (stage1/model.py:342-433, vqgan.py:84-116, quantize.py:271-312)
    bool has_vq_enc = false;
    int enc_cin_pad = 0;
    ConvW enc_conv_in, enc_conv_out, quant_conv;
    std::vector<DownLevelW> down;
    ResBlockW enc_mid1, enc_mid2;
    AttnBlockW enc_mid_attn;
    const float *enc_norm_out_w = nullptr, *enc_norm_out_b = nullptr;
    float* codebook_sqnorm = nullptr;

    // ---- split-precision mode (BEVGEN_PRECISION_F16X3): (hi, lo) f16 planes of every GEMM / conv weight, keyed by its fp32 device pointer
    std::unordered_map<const float*, SplitPlanes> split;
    void split_weight(const float* w, long n);

    Profiler prof;   // per-launch HIP-event timing (bevgen_profile_begin / _end) of THIS context

    // ---- hipGraph replay of the Route A decode step
    hipStream_t graph_stream = nullptr;
    hipEvent_t graph_ev_in = nullptr, graph_ev_out = nullptr;
    // side streams of the chain-pipelined decode step (chains 1..n-1; chain 0 runs on the step's own stream) and their fork / join events
    std::vector<hipStream_t> chain_streams;
    hipEvent_t chain_fork = nullptr;
    std::vector<hipEvent_t> chain_join;
    std::vector<std::pair<hipGraphExec_t, hipGraph_t>> retired_graphs;
    // the instantiated decode-step graph is kept and replayed by later bevgen_ar_sample calls that bake in the same pointers / parameters
    struct GraphKey {
        int B = 0, G = 0, top_k = 0, greedy = 0, kv = 0, chains = 0; float temperature = 0.f;
        const void *noise = nullptr, *forced = nullptr, *out = nullptr, *arena = nullptr, *persist = nullptr, *trace = nullptr;
        bool operator==(const GraphKey& o) const {
            return B == o.B && G == o.G && chains == o.chains && top_k == o.top_k && greedy == o.greedy && kv == o.kv && temperature == o.temperature && noise == o.noise &&
                   forced == o.forced && out == o.out && arena == o.arena && persist == o.persist && trace == o.trace;
        }
    } graph_key;
    hipGraphExec_t graph_exec = nullptr;
    hipGraph_t graph = nullptr;
    // per-step timing of the graph replays (bevgen_ar_step_timing): one event after every replay
    bool time_steps = false;
    std::vector<hipEvent_t> step_events;
    int step_events_used = 0;
    bool disable_graphs = false;
    // experiment switch ($BEVGEN_PREFETCH = 1 both MLP images, 2 the up-projection only): the decode attention kernels pull the layer's MLP weight images through L2
    // under the tail of their K/V walk.  Measured slower on the same box (1.55 -> 1.65 / 1.61 ms/step, profiles/r03_ab_prefetch.txt): default off
    // fused MLP launch of the decode step (decode_fused.hip ar_mlp_fused_kernel): barrier words (device, zeroed at finalize) and the host-visible error word
    unsigned* mlpf_sync = nullptr;
    bool mlpf_disabled = false;   // a fused MLP launch of this context reported a timeout / wrong placement: every later step takes the two-launch form
    // device status word (common.h BG_ST_*): one mapped host word per context, allocated at bevgen_create; kernels OR bits into it, the host reads it without synchronising
    unsigned *status_host = nullptr, *status_dev = nullptr;
    void check_status(const char* when);   // throws (and clears the word) when a kernel of an earlier / the synchronised call raised a bit
    long long* trace = nullptr;   // diagnostics (bevgen_set_trace_buffer): phase timestamps of the fused decode kernels, [3 kinds][4096 workgroups][8]
    void retire_graph(hipGraphExec_t e, hipGraph_t g);  // destroyed once the stream has drained (next call or destroy)

    ~Ctx();
    const DevTensor& need(const std::string& name) const;
    const DevTensor* find(const std::string& name) const;
    const float* pf(const std::string& name) const { return need(name).f(); }
    void* own(size_t bytes);  // hipMalloc tracked for destroy
};

// context.cpp
void ctx_load_tensor(Ctx& c, const char* name, const void* h, int dtype, int ndim, const int64_t* shape);
void ctx_finalize(Ctx& c);
void ctx_pack_split_qkv(Ctx& c, hipStream_t s);   // decode_path = auto: QKV operand image of the split decode layer, packed on first need
// muse.cpp
void muse_forward(Ctx& c, const int64_t* ids, const int64_t* cond, const float* I_inv, const float* E_inv, int B, float* logits, float* embed, hipStream_t s);
void maskgit_generate(Ctx& c, const int64_t* cond, const float* I_inv, const float* E_inv, int B, int timesteps, const int32_t* sched, float temperature,
                      int topk_k, float critic_noise_scale, const float* gumbel_u, const float* critic_u, const int64_t* init_ids, int64_t* out, hipStream_t s,
                      unsigned long long noise_seed = 0, int score_mode = 0 /* 0 token critic, 1 / 2 softmax confidence (muse_net:611-622) */, int samples_per_layout = 1);
// ar.cpp
void sparse_self_attention_op(Ctx& c, const float* q, const float* k, const float* v, const int64_t* layout, const float* mask, const float* add, int B, int H,
                              int L, int block, float* out, hipStream_t s);
void ar_prefill(Ctx& c, const int64_t* cond, const float* I_inv, const float* E_inv, int B, hipStream_t s, int samples_per_layout = 1);
void ar_logits(Ctx& c, float* logits, hipStream_t s);
void ar_decode_step(Ctx& c, const int64_t* tok, hipStream_t s);
void ar_sample(Ctx& c, const int64_t* cond, const float* I_inv, const float* E_inv, int B, int steps, int top_k, float temperature, int greedy,
               const float* noise_u, int samples_per_layout, const int64_t* forced, int64_t* out, float* step_logits, hipStream_t s);
int ar_step_times(Ctx& c, float* out_ms, int cap);
// vqdec.cpp
void vq_finalize(Ctx& c);
void vq_decode(Ctx& c, const int64_t* ids, const float* latents_nchw, int n, int lat_h, int lat_w, int out_mode, void* out, hipStream_t s);
void vq_encode(Ctx& c, const float* x_nchw, int n, int RH, int RW, int64_t* ids, hipStream_t s);
// vqenc_kernels.hip
void launch_nchw_to_nhwc_pad(const float* x, float* y, int n, int hw, int C, int Cpad, hipStream_t s);
void launch_relayout_conv_weight_pad(const float* w, float* o, int cout, int cin, int cin_pad, int kh, int kw, hipStream_t s);
void launch_row_sqnorm(const float* x, float* out, long rows, int D, hipStream_t s);
void launch_vq_argmin(const float* dots, const float* zz, const float* ee, int64_t* ids, long rows, int n_e, hipStream_t s);
// tables.hip
void launch_build_attn_bias(const float* tril_emb /*or null*/, const float* prob /*or null*/, float* out, int L, hipStream_t s);
void launch_build_muse_bias(const float* attn_bias, int L, int K, int N, float* bias_self, int ldS, float* bias_cross, int ldC, hipStream_t s);
void launch_build_allowed(const float* mask, uint8_t* out, long n, hipStream_t s);
void launch_build_layout(const int64_t* layout, uint8_t* lay, uint16_t* chunks, int heads, int nb, int blk, int L, int chunks_ld, hipStream_t s);
void launch_build_masked_bias(const float* add /*[L,L] or null*/, const SparseVis& vis, float* out, int heads, int rows, int cols, int ldout, int ldadd, float scale,
                              hipStream_t s);
void launch_ar_step_embed(const int64_t* tok, const float* tok_emb, const float* img_embed, const float* pos_emb, const int64_t* fwd_idx, const int* d_step,
                          float* x, int B, int C, int T, int D, int vocab_rows, hipStream_t s);
void launch_store_tokens(const int64_t* tok, const int64_t* fwd_idx, const int* d_step, int64_t* out, int B, int N, hipStream_t s);
void launch_gather_rows(const float* x, float* out, int B, int row, int rows_per_batch, int D, hipStream_t s);
void launch_replicate_prefix(void* kc, void* vc, int layers, int B, int H, int L, int rows, int src, int dst0, int count, int elem_bytes, hipStream_t s);
void launch_increment(int* p, hipStream_t s);
void launch_geglu_weight_order(const float* W, float* Wo, int F, int Fpad, int D, hipStream_t s);
void launch_round_to_f16(float* w /* rounded in place */, void* h /* fp16 copy or null */, long n, hipStream_t s);
void launch_relayout_conv_weight(const float* w_oihw, float* w_ohwi, int cout, int cin, int kh, int kw, hipStream_t s);
void launch_fuse_qkv(const float* wq, const float* wk, const float* wv, const float* bq, const float* bk, const float* bv, float* w, float* b, int D, hipStream_t s);
void launch_pad_rows(const float* src, int ld_src, float* dst, int ld_dst, int rows, int cols, hipStream_t s);
void launch_fill_i64(int64_t* p, long n, int64_t v, hipStream_t s);

}  // namespace bevgen
