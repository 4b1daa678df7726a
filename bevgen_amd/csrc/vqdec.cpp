// Stage-1 VQGAN decode:  ids -> codebook rows -> post_quant_conv -> Decoder -> (denormalise)
//   VectorQuantizer2.get_codebook_entry   stage1/quantize.py:314-329
//   VQModel.decode                        stage1/vqgan.py:118-121
//   Decoder.forward                       stage1/model.py:506-537   ResnetBlock :117-137, AttnBlock :168-192, Upsample :49-53
//   util.denormalize_tensor               bev_utils/util.py:97-118
//
// Activations are NHWC so that every convolution is an implicit GEMM with K = (tap, channel) contiguous:
//   3x3 conv  -> gemm.hip MODE_CONV3 (im2col gather in the A-tile loader, nearest-2x upsample fused into the gather)
//   1x1 conv  -> plain GEMM over pixels
//   AttnBlock -> four 1x1 GEMMs + batched Q K^T / softmax / P V GEMMs (single head of width C, 256 tokens)
// GroupNorm(32)+swish is a stats pass + an elementwise pass; residual adds ride in the GEMM epilogues.
#include "model.h"

namespace bevgen {

namespace {

const std::string kPrefix = "first_stage_model.";

ConvW load_conv(Ctx& c, const std::string& name, int k) {
    const DevTensor& w = c.need(kPrefix + name + ".weight");
    BG_REQUIRE(w.shape.size() == 4 && w.shape[2] == k && w.shape[3] == k, "conv '%s' is not a %dx%d kernel", name.c_str(), k, k);
    ConvW cw;
    cw.cout = (int)w.shape[0];
    cw.cin = (int)w.shape[1];
    cw.k = k;
    cw.b = c.pf(kPrefix + name + ".bias");
    cw.w = reinterpret_cast<float*>(c.own(w.bytes));
    launch_relayout_conv_weight(w.f(), cw.w, cw.cout, cw.cin, k, k, 0);
    c.split_weight(cw.w, (long)cw.cout * cw.cin * k * k);
    return cw;
}

ResBlockW load_res(Ctx& c, const std::string& p) {
    ResBlockW r;
    r.n1w = c.pf(kPrefix + p + "norm1.weight"); r.n1b = c.pf(kPrefix + p + "norm1.bias");
    r.n2w = c.pf(kPrefix + p + "norm2.weight"); r.n2b = c.pf(kPrefix + p + "norm2.bias");
    r.c1 = load_conv(c, p + "conv1", 3);
    r.c2 = load_conv(c, p + "conv2", 3);
    r.cin = r.c1.cin; r.cout = r.c1.cout;
    r.has_nin = c.find(kPrefix + p + "nin_shortcut.weight") != nullptr;
    if (r.has_nin) r.nin = load_conv(c, p + "nin_shortcut", 1);
    BG_REQUIRE(r.has_nin == (r.cin != r.cout), "resnet block '%s': shortcut/channels mismatch", p.c_str());
    return r;
}

AttnBlockW load_attn(Ctx& c, const std::string& p) {
    AttnBlockW a;
    a.nw = c.pf(kPrefix + p + "norm.weight"); a.nb = c.pf(kPrefix + p + "norm.bias");
    a.q = load_conv(c, p + "q", 1); a.k = load_conv(c, p + "k", 1); a.v = load_conv(c, p + "v", 1); a.proj = load_conv(c, p + "proj_out", 1);
    a.c = a.q.cin;
    return a;
}

struct Act { float* p; int n, h, w, c; long elems() const { return (long)n * h * w * c; } };

// GroupNorm statistics of a tensor as partial sums out of the epilogue of the convolution that produced it (GemmArgs::gn_part): `of` names the tensor they describe.
// Every writer of an activation buffer goes through note_write() so that a stale association can never be used.
struct GnPart { float* buf = nullptr; const float* of = nullptr; };
static void note_write(GnPart* gp, const float* y) { if (gp && gp->of == y) gp->of = nullptr; }

// planes: x.p holds the interleaved (hi, lo) f16 plane image of the activation (written by gn(..., planes = true)) instead of fp32
void conv3(const Act& x, const ConvW& w, float* y, const float* residual, int up, hipStream_t s, bool planes = false, GnPart* gp = nullptr) {
    GemmArgs g;
    note_write(gp, y);
    const int oh = up ? x.h * 2 : x.h, ow = up ? x.w * 2 : x.w;
    g.mode = MODE_CONV3;
    if (planes) { g.A_hi = reinterpret_cast<const uint16_t*>(x.p); g.A_lo = g.A_hi + 32; }
    else g.A = x.p;
    g.B = w.w; g.C = y; g.R = residual; g.bias_n = w.b;
    g.M = x.n * oh * ow; g.N = w.cout; g.K = 9 * w.cin;
    g.lda = w.cin; g.ldb = 9 * w.cin; g.ldc = w.cout; g.ldr = w.cout;
    g.conv_h = oh; g.conv_w = ow; g.conv_cin = w.cin; g.conv_up = up;
    // the LDS-DMA kernel (plane input) also leaves the GroupNorm partial sums of its output where the shape allows: the consumer's statistics pass disappears
    static const int gn_epi = getenv("BEVGEN_GN_EPILOGUE") ? atoi(getenv("BEVGEN_GN_EPILOGUE")) : 1;   // (0: always the statistics pass, for A/B runs)
    if (gn_epi && gp && gp->buf && planes && groupnorm_partials_supported(oh * ow, w.cout)) { g.gn_part = gp->buf; gp->of = y; }
    launch_gemm(g, s);
}

void conv1(const float* x, long rows, const ConvW& w, float* y, const float* residual, hipStream_t s, GnPart* gp = nullptr) {
    note_write(gp, y);
    GemmArgs g;
    g.A = x; g.B = w.w; g.C = y; g.R = residual; g.bias_n = w.b;
    g.M = (int)rows; g.N = w.cout; g.K = w.cin;
    g.lda = w.cin; g.ldb = w.cin; g.ldc = w.cout; g.ldr = w.cout;
    launch_gemm(g, s);
}

struct DecWs {
    float *a, *b, *t;   // ping-pong activations + temp (each max_act floats)
    float* stats;       // [n*32*2]
    void* gn_ws;
    float *q, *k, *vT, *S;
    GnPart part;        // epilogue partials of the most recent convolution output (buf: groupnorm_part_floats of the widest level, or null)
};

// statistics of x: from the producing convolution's epilogue partials when they describe exactly this tensor, else the pass over the tensor
void gn_stats(const Act& x, DecWs& ws, hipStream_t s) {
    if (ws.part.of == x.p && ws.part.buf) launch_groupnorm_stats_from_partials(ws.part.buf, ws.stats, x.n, x.h * x.w, x.c, 1e-6f, s);
    else launch_groupnorm_stats(x.p, ws.stats, ws.gn_ws, x.n, x.h * x.w, x.c, 1e-6f, s);
}

void gn(const Act& x, const float* w, const float* b, float* y, int swish, DecWs& ws, hipStream_t s, bool planes = false) {
    gn_stats(x, ws, s);
    note_write(&ws.part, y);
    if (planes) launch_groupnorm_apply_planes(x.p, ws.stats, w, b, y, x.n, x.h * x.w, x.c, swish, s);
    else launch_groupnorm_apply(x.p, ws.stats, w, b, y, x.n, x.h * x.w, x.c, swish, s);
}

// x (in ws.a-or-b) -> out buffer `y`; uses ws.t and the other ping-pong buffer as scratch
void resblock(const ResBlockW& r, Act& x, float* y, float* scratch, DecWs& ws, hipStream_t s, bool planes) {
    // h = conv1(swish(norm1(x)));  split-precision mode: the normalised activation is written directly as the (hi, lo) planes the convolution reads
    gn(x, r.n1w, r.n1b, ws.t, 1, ws, s, planes);
    Act t{ws.t, x.n, x.h, x.w, r.cin};
    conv3(t, r.c1, scratch, nullptr, 0, s, planes, &ws.part);
    Act h1{scratch, x.n, x.h, x.w, r.cout};
    gn(h1, r.n2w, r.n2b, ws.t, 1, ws, s, planes);
    Act t2{ws.t, x.n, x.h, x.w, r.cout};
    const float* shortcut = x.p;
    if (r.has_nin) {  // x = nin_shortcut(x)  (1x1), written over h1 (no longer needed after norm2)
        conv1(x.p, (long)x.n * x.h * x.w, r.nin, scratch, nullptr, s, &ws.part);
        shortcut = scratch;
    }
    conv3(t2, r.c2, y, shortcut, 0, s, planes, &ws.part);  // y = x + conv2(...); its epilogue also leaves the statistics the next block's norm1 needs
    x = Act{y, x.n, x.h, x.w, r.cout};
}

void attnblock(const AttnBlockW& a, Act& x, float* y, DecWs& ws, hipStream_t s) {
    const int hw = x.h * x.w, C = a.c, n = x.n;
    const int hwp = (int)round_up(hw, 32);   // key dimension padded to the GEMM's k granularity (non-square latents: 14 x 25 = 350 -> 352); pad keys carry P = 0, v = 0
    const long rows = (long)n * hw;
    gn(x, a.nw, a.nb, ws.t, 0, ws, s);
    conv1(ws.t, rows, a.q, ws.q, nullptr, s);
    conv1(ws.t, rows, a.k, ws.k, nullptr, s);
    if (hwp != hw) HIP_CHECK(hipMemsetAsync(ws.vT, 0, (size_t)n * C * hwp * sizeof(float), s));
    {   // vT[img] [C, hw] = Wv [C,C] * h_img^T + bias (per row)
        GemmArgs g;
        g.A = a.v.w; g.B = ws.t; g.C = ws.vT; g.bias_m = a.v.b;
        g.M = C; g.N = hw; g.K = C; g.lda = C; g.ldb = C; g.ldc = hwp;
        g.batch = n; g.strideA = 0; g.strideB = (long)hw * C; g.strideC = (long)C * hwp;
        launch_gemm(g, s);
    }
    {   // S[img] [hw, hw] = q k^T
        GemmArgs g;
        g.A = ws.q; g.B = ws.k; g.C = ws.S;
        g.M = hw; g.N = hw; g.K = C; g.lda = C; g.ldb = C; g.ldc = hwp;
        g.batch = n; g.strideA = (long)hw * C; g.strideB = (long)hw * C; g.strideC = (long)hw * hwp;
        launch_gemm(g, s);
    }
    launch_row_softmax(ws.S, (int)rows, hw, 1.0f / sqrtf((float)C), s, hwp);  // w_ * int(c)**-0.5, softmax over keys
    {   // O[img] [hw, C] = P [hw,hw] * v [hw, C]  (B operand = vT [C, hw])
        GemmArgs g;
        g.A = ws.S; g.B = ws.vT; g.C = ws.q;  // reuse q as the output buffer
        g.M = hw; g.N = C; g.K = hwp; g.lda = hwp; g.ldb = hwp; g.ldc = C;
        g.batch = n; g.strideA = (long)hw * hwp; g.strideB = (long)C * hwp; g.strideC = (long)hw * C;
        launch_gemm(g, s);
    }
    conv1(ws.q, rows, a.proj, y, x.p, s, &ws.part);  // x + proj_out(h)
    x = Act{y, n, x.h, x.w, C};
}

}  // namespace

static void vq_enc_finalize(Ctx& c);

void vq_finalize(Ctx& c) {
    const auto& g = c.cfg;
    BG_REQUIRE(g.vq_num_levels >= 1 && g.vq_num_levels <= 8, "vq_num_levels out of range");
    const DevTensor& cb = c.need(kPrefix + "quantize.embedding.weight");
    BG_REQUIRE(cb.shape.size() == 2 && cb.shape[0] == g.vq_n_embed && cb.shape[1] == g.vq_embed_dim, "codebook must be [n_embed, embed_dim]");
    c.codebook = cb.f();
    c.has_vq = c.find(kPrefix + "decoder.conv_in.weight") != nullptr;
    c.has_vq_enc = c.find(kPrefix + "encoder.conv_in.weight") != nullptr;
    BG_REQUIRE(c.has_vq || c.has_vq_enc, "VQGAN context: neither decoder.* nor encoder.* tensors were loaded");
    if (c.has_vq_enc) vq_enc_finalize(c);
    if (!c.has_vq) return;
    c.post_quant = load_conv(c, "post_quant_conv", 1);
    c.conv_in = load_conv(c, "decoder.conv_in", 3);
    c.mid1 = load_res(c, "decoder.mid.block_1.");
    c.mid_attn = load_attn(c, "decoder.mid.attn_1.");
    c.mid2 = load_res(c, "decoder.mid.block_2.");
    c.up.clear();
    c.up.resize(g.vq_num_levels);
    int res = g.vq_resolution >> (g.vq_num_levels - 1);
    for (int lvl = g.vq_num_levels - 1; lvl >= 0; --lvl) {
        UpLevelW& u = c.up[lvl];
        for (int b = 0; b < g.vq_num_res_blocks + 1; ++b) {
            const std::string p = "decoder.up." + std::to_string(lvl) + ".";
            u.blocks.push_back(load_res(c, p + "block." + std::to_string(b) + "."));
            if (res == g.vq_attn_resolution) u.attns.push_back(load_attn(c, p + "attn." + std::to_string(b) + "."));
        }
        u.has_up = lvl != 0;
        if (u.has_up) {
            u.up = load_conv(c, "decoder.up." + std::to_string(lvl) + ".upsample.conv", 3);
            res *= 2;
        }
    }
    c.norm_out_w = c.pf(kPrefix + "decoder.norm_out.weight");
    c.norm_out_b = c.pf(kPrefix + "decoder.norm_out.bias");
    c.conv_out = load_conv(c, "decoder.conv_out", 3);
    const float mean[3] = {0.4265f, 0.4489f, 0.4769f}, stdv[3] = {0.2053f, 0.2206f, 0.2578f};
    c.denorm_mean = reinterpret_cast<float*>(c.own(sizeof mean));
    c.denorm_std = reinterpret_cast<float*>(c.own(sizeof stdv));
    HIP_CHECK(hipMemcpy(c.denorm_mean, mean, sizeof mean, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(c.denorm_std, stdv, sizeof stdv, hipMemcpyHostToDevice));
}

// The decoder is fully convolutional (stage1/model.py:506-537): the latent grid is lat_h x lat_w (cam_latent_res, e.g. 16 x 16 or nuScenes' 14 x 25), the output
// (lat_h << (levels-1)) x (lat_w << (levels-1)); which levels carry AttnBlocks is decided by ddconfig.resolution alone (curr_res == attn_resolutions, :466-480).
// out_mode: 0 = raw fp32, 1 = denormalised fp32 in [0,1], 2 = denormalised uint8 round(x*255) (`out` then points to uint8).
void vq_decode(Ctx& c, const int64_t* ids, const float* latents_nchw, int n_total, int lat_h, int lat_w, int out_mode, void* out, hipStream_t s) {
    BG_REQUIRE(c.has_vq, "this context holds no VQGAN decoder weights (decoder.* tensors were not loaded)");
    const auto& g = c.cfg;
    const bool planes = g.precision == BEVGEN_PRECISION_F16X3;   // split-precision mode: GroupNorm writes (hi, lo) planes, convolutions read them by LDS-DMA
    BG_REQUIRE(lat_h >= 1 && lat_w >= 1, "vq_decode: latent grid %d x %d", lat_h, lat_w);
    const int denorm = out_mode != 0;
    const long lat_hw = (long)lat_h * lat_w;
    const int RH = lat_h << (g.vq_num_levels - 1), RW = lat_w << (g.vq_num_levels - 1);
    BG_REQUIRE(!denorm || g.vq_out_ch == 3, "denormalize needs 3 output channels");
    // widest activation per image: max over levels of h*w * channels; most attention tokens over the levels that carry AttnBlocks
    long per_img = 0, attn_hw = lat_hw;
    int max_c = 0;
    {
        long hw = lat_hw;
        for (int lvl = g.vq_num_levels - 1; lvl >= 0; --lvl) {
            const int ch = g.vq_ch * g.vq_ch_mult[lvl];
            const int ch_in = lvl == g.vq_num_levels - 1 ? ch : g.vq_ch * g.vq_ch_mult[lvl + 1];
            per_img = std::max<long>(per_img, hw * std::max(ch, ch_in));
            max_c = std::max(max_c, std::max(ch, ch_in));
            if (!c.up[lvl].attns.empty()) attn_hw = std::max(attn_hw, hw);
            if (lvl != 0) hw *= 4;
        }
    }
    const long attn_hwp = round_up(attn_hw, 32);
    // images per pass: the 16 x 16 stages only fill the chip (256 x 128 tiles on 256 CUs) from ~48 images; 148 -> 128 ms per 96 images vs 16, arena ~10 GB.  (Round 5: smaller
    // passes whose full-resolution tensors would fit the 256 MB memory-side cache between conv -> statistics -> apply -> conv are slower throughout: 6 / 12 / 24 images per
    // pass 11.96 / 9.14 / 7.96 ms per scene against 7.47, profiles/r05_ab_vq.txt)
    const int chunk_max = 48;
    const int attn_c = max_c;
    const int chunk = std::min(n_total, chunk_max);
    const size_t act_b = (size_t)per_img * chunk * sizeof(float);
    const size_t need = 4 * act_b + act_b / 64 + (size_t)chunk * 64 * sizeof(float) + groupnorm_ws_bytes(chunk, RH * RW) +
                        (size_t)chunk * attn_hwp * (3 * attn_c + attn_hwp) * sizeof(float) + (size_t)chunk * lat_hw * g.vq_embed_dim * sizeof(float) +
                        (size_t)chunk * RH * RW * 4 * sizeof(float) + 34 * 256;
    c.arena.reserve(need);
    for (int i0 = 0; i0 < n_total; i0 += chunk) {
        const int n = std::min(chunk, n_total - i0);
        c.arena.reset();
        DecWs ws;
        if (planes) ws.part.buf = c.arena.get<float>((size_t)per_img * n / 64 + 64);   // GroupNorm partials of one tensor: (pixels / 32) x (channels / 4) x 2
        ws.a = c.arena.get<float>((size_t)per_img * n);
        ws.b = c.arena.get<float>((size_t)per_img * n);
        ws.t = c.arena.get<float>((size_t)per_img * n);
        ws.stats = c.arena.get<float>((size_t)n * 64);
        ws.gn_ws = c.arena.alloc(groupnorm_ws_bytes(n, RH * RW));
        ws.q = c.arena.get<float>((size_t)n * attn_hwp * attn_c);
        ws.k = c.arena.get<float>((size_t)n * attn_hwp * attn_c);
        ws.vT = c.arena.get<float>((size_t)n * attn_hwp * attn_c);
        ws.S = c.arena.get<float>((size_t)n * attn_hwp * attn_hwp);
        float* zq = c.arena.get<float>((size_t)n * lat_hw * g.vq_embed_dim);
        float* img = c.arena.get<float>((size_t)n * RH * RW * 4);

        const long lrows = (long)n * lat_hw;
        if (ids) launch_codebook_gather(ids + (long)i0 * lat_hw, c.codebook, zq, (int)lrows, g.vq_embed_dim, g.vq_n_embed, s);
        else launch_nchw_to_nhwc(latents_nchw + (long)i0 * g.vq_embed_dim * lat_hw, zq, n, (int)lat_hw, g.vq_embed_dim, s);
        conv1(zq, lrows, c.post_quant, ws.t, nullptr, s);                       // post_quant_conv
        Act x{ws.t, n, lat_h, lat_w, g.vq_z_channels};
        conv3(x, c.conv_in, ws.a, nullptr, 0, s, false, &ws.part);               // conv_in
        x = Act{ws.a, n, lat_h, lat_w, c.conv_in.cout};
        // three rotating activation buffers (input / scratch / output of a block) + ws.t for the normalised tensor
        float* o = c.arena.get<float>((size_t)per_img * n);
        auto res_step = [&](const ResBlockW& r) {
            // buffers: x.p in {a,b,o}; pick scratch and y as the two others
            float* bufs[3] = {ws.a, ws.b, o};
            float* scratch = nullptr; float* y = nullptr;
            for (float* b : bufs) { if (b != x.p) { if (!scratch) scratch = b; else y = b; } }
            resblock(r, x, y, scratch, ws, s, planes);
        };
        auto attn_step = [&](const AttnBlockW& ab) {
            float* bufs[3] = {ws.a, ws.b, o};
            float* y = nullptr;
            for (float* b : bufs) if (b != x.p) { y = b; break; }
            attnblock(ab, x, y, ws, s);
        };
        res_step(c.mid1);
        attn_step(c.mid_attn);
        res_step(c.mid2);
        for (int lvl = g.vq_num_levels - 1; lvl >= 0; --lvl) {
            const UpLevelW& u = c.up[lvl];
            for (size_t b = 0; b < u.blocks.size(); ++b) {
                res_step(u.blocks[b]);
                if (!u.attns.empty()) attn_step(u.attns[b]);
            }
            if (u.has_up) {  // nearest 2x + 3x3 conv, fused
                float* bufs[3] = {ws.a, ws.b, o};
                float* y = nullptr;
                for (float* b : bufs) if (b != x.p) { y = b; break; }
                // The upsample convolution reads the block output directly (no GroupNorm in front of it, s1model:49-53): in split-precision mode the tensor is first
                // re-laid out as (hi, lo) planes (one elementwise pass over the SMALL pre-upsample tensor) so that the 4x larger convolution runs on the LDS-DMA kernel
                // instead of the register-staged one ($BEVGEN_VQ_UP_PLANES=0: as in rounds 2-4)
                static const int up_planes = getenv("BEVGEN_VQ_UP_PLANES") ? atoi(getenv("BEVGEN_VQ_UP_PLANES")) : 1;
                const int q4 = x.c >> 2;
                if (planes && up_planes && (q4 & (q4 - 1)) == 0 && x.c % 32 == 0) {
                    note_write(&ws.part, ws.t);
                    launch_to_planes(x.p, ws.t, x.n, x.h * x.w, x.c, s);
                    Act xp{ws.t, x.n, x.h, x.w, x.c};
                    conv3(xp, u.up, y, nullptr, 1, s, true, &ws.part);
                } else {
                    conv3(x, u.up, y, nullptr, 1, s, false, &ws.part);
                }
                x = Act{y, x.n, x.h * 2, x.w * 2, u.up.cout};
            }
        }
        const long o_off = (long)i0 * g.vq_out_ch * RH * RW;
        static const bool tail_off = getenv("BEVGEN_VQ_TAIL") && atoi(getenv("BEVGEN_VQ_TAIL")) == 0;   // (A/B switch: 0 = the three-kernel tail)
        if (!tail_off && vq_out_conv_supported(x.c, g.vq_out_ch)) {   // norm_out + swish + conv_out + denormalise + layout in one kernel
            gn_stats(x, ws, s);
            launch_vq_out_conv(x.p, ws.stats, c.norm_out_w, c.norm_out_b, c.conv_out.w, c.conv_out.b, denorm ? c.denorm_mean : nullptr, denorm ? c.denorm_std : nullptr, denorm ? 1 : 0,
                               out_mode == 2 ? nullptr : reinterpret_cast<float*>(out) + o_off, out_mode == 2 ? reinterpret_cast<uint8_t*>(out) + o_off : nullptr, x.n, x.h, x.w, x.c,
                               g.vq_out_ch, s);
            continue;
        }
        gn(x, c.norm_out_w, c.norm_out_b, ws.t, 1, ws, s, planes);
        Act t{ws.t, x.n, x.h, x.w, x.c};
        conv3(t, c.conv_out, img, nullptr, 0, s, planes);  // [n, RH*RW, out_ch]
        launch_nhwc_to_nchw(img, out_mode == 2 ? nullptr : reinterpret_cast<float*>(out) + o_off, n, RH * RW, g.vq_out_ch, g.vq_out_ch, denorm ? c.denorm_mean : nullptr,
                            denorm ? c.denorm_std : nullptr, denorm ? 1 : 0, s, out_mode == 2 ? reinterpret_cast<uint8_t*>(out) + o_off : nullptr);
    }
}

// =====================================================================================================
// Encoder + quantizer: the step BEFORE the sampling path (encode_to_c / encode_to_z, muse_lm:142-155)
//   Encoder.forward        stage1/model.py:405-433   (Downsample :56-75: zero pad (0,1,0,1) then 3x3 stride-2 conv)
//   VQModel.encode         stage1/vqgan.py:84-116    (geometric_embedding=False in the released configs)
//   VectorQuantizer2.forward  stage1/quantize.py:271-312
// =====================================================================================================
static ConvW load_conv_in_padded(Ctx& c, const std::string& name, int cin_pad) {
    const DevTensor& w = c.need(kPrefix + name + ".weight");
    BG_REQUIRE(w.shape.size() == 4 && w.shape[2] == 3 && w.shape[3] == 3, "conv '%s' is not a 3x3 kernel", name.c_str());
    ConvW cw;
    cw.cout = (int)w.shape[0];
    cw.cin = cin_pad;
    cw.k = 3;
    cw.b = c.pf(kPrefix + name + ".bias");
    cw.w = reinterpret_cast<float*>(c.own((size_t)cw.cout * 9 * cin_pad * sizeof(float)));
    launch_relayout_conv_weight_pad(w.f(), cw.w, cw.cout, (int)w.shape[1], cin_pad, 3, 3, 0);
    c.split_weight(cw.w, (long)cw.cout * 9 * cin_pad);
    return cw;
}

static void vq_enc_finalize(Ctx& c) {
    const auto& g = c.cfg;
    BG_REQUIRE(g.vq_in_channels >= 1, "vq_in_channels must be set for the encoder");
    c.enc_cin_pad = (int)round_up(g.vq_in_channels, 32);
    c.enc_conv_in = load_conv_in_padded(c, "encoder.conv_in", c.enc_cin_pad);
    c.down.clear();
    c.down.resize(g.vq_num_levels);
    int res = g.vq_resolution;
    for (int lvl = 0; lvl < g.vq_num_levels; ++lvl) {
        DownLevelW& d = c.down[lvl];
        const std::string p = "encoder.down." + std::to_string(lvl) + ".";
        for (int b = 0; b < g.vq_num_res_blocks; ++b) {
            d.blocks.push_back(load_res(c, p + "block." + std::to_string(b) + "."));
            if (res == g.vq_attn_resolution) d.attns.push_back(load_attn(c, p + "attn." + std::to_string(b) + "."));
        }
        d.has_down = lvl != g.vq_num_levels - 1;
        if (d.has_down) {
            d.down = load_conv(c, p + "downsample.conv", 3);
            res /= 2;
        }
    }
    c.enc_mid1 = load_res(c, "encoder.mid.block_1.");
    c.enc_mid_attn = load_attn(c, "encoder.mid.attn_1.");
    c.enc_mid2 = load_res(c, "encoder.mid.block_2.");
    c.enc_norm_out_w = c.pf(kPrefix + "encoder.norm_out.weight");
    c.enc_norm_out_b = c.pf(kPrefix + "encoder.norm_out.bias");
    c.enc_conv_out = load_conv(c, "encoder.conv_out", 3);
    c.quant_conv = load_conv(c, "quant_conv", 1);
    BG_REQUIRE(c.enc_conv_out.cout == g.vq_z_channels && c.quant_conv.cout == g.vq_embed_dim, "encoder output channels do not match z_channels / embed_dim (double_z must be False)");
    c.codebook_sqnorm = reinterpret_cast<float*>(c.own((size_t)g.vq_n_embed * sizeof(float)));
    launch_row_sqnorm(c.codebook, c.codebook_sqnorm, g.vq_n_embed, g.vq_embed_dim, 0);
}

// x [n, in_channels, RH, RW] with RH, RW multiples of 2^(levels-1) (fully convolutional: 256 x 256 or nuScenes' 224 x 400) -> ids [n, (RH >> (levels-1)) * (RW >> (levels-1))]
void vq_encode(Ctx& c, const float* x_nchw, int n_total, int RH, int RW, int64_t* ids, hipStream_t s) {
    BG_REQUIRE(c.has_vq_enc, "this context holds no VQGAN encoder weights (encoder.* tensors were not loaded)");
    const auto& g = c.cfg;
    const bool planes = g.precision == BEVGEN_PRECISION_F16X3;   // split-precision mode: GroupNorm writes (hi, lo) planes, convolutions read them by LDS-DMA
    const int f = 1 << (g.vq_num_levels - 1);
    BG_REQUIRE(RH >= f && RW >= f && RH % f == 0 && RW % f == 0, "vq_encode: image %d x %d is not a multiple of the downsampling factor %d", RH, RW, f);
    const int lat_h = RH / f, lat_w = RW / f;
    const long lat_hw = (long)lat_h * lat_w;
    long per_img = (long)RH * RW * std::max(c.enc_cin_pad, g.vq_ch), attn_hw = lat_hw;
    int max_c = g.vq_ch;
    {
        long hw = (long)RH * RW;
        for (int lvl = 0; lvl < g.vq_num_levels; ++lvl) {
            const int ch = g.vq_ch * g.vq_ch_mult[lvl];
            per_img = std::max<long>(per_img, hw * ch);
            max_c = std::max(max_c, ch);
            if (!c.down[lvl].attns.empty()) attn_hw = std::max(attn_hw, hw);
            if (lvl != g.vq_num_levels - 1) hw /= 4;
        }
    }
    const int chunk = std::min(n_total, 48);
    const long attn_hwp = round_up(attn_hw, 32);
    const size_t act_b = (size_t)per_img * chunk * sizeof(float);
    const size_t need = 4 * act_b + (size_t)chunk * 64 * sizeof(float) + groupnorm_ws_bytes(chunk, RH * RW) +
                        (size_t)chunk * attn_hwp * (3 * max_c + attn_hwp) * sizeof(float) + (size_t)chunk * lat_hw * (g.vq_n_embed + g.vq_embed_dim + 1) * sizeof(float) + 32 * 256;
    c.arena.reserve(need);
    for (int i0 = 0; i0 < n_total; i0 += chunk) {
        const int n = std::min(chunk, n_total - i0);
        c.arena.reset();
        DecWs ws;
        ws.a = c.arena.get<float>((size_t)per_img * n);
        ws.b = c.arena.get<float>((size_t)per_img * n);
        ws.t = c.arena.get<float>((size_t)per_img * n);
        float* o = c.arena.get<float>((size_t)per_img * n);
        ws.stats = c.arena.get<float>((size_t)n * 64);
        ws.gn_ws = c.arena.alloc(groupnorm_ws_bytes(n, RH * RW));
        ws.q = c.arena.get<float>((size_t)n * attn_hwp * max_c);
        ws.k = c.arena.get<float>((size_t)n * attn_hwp * max_c);
        ws.vT = c.arena.get<float>((size_t)n * attn_hwp * max_c);
        ws.S = c.arena.get<float>((size_t)n * attn_hwp * attn_hwp);
        const long lrows = (long)n * lat_hw;
        float* dots = c.arena.get<float>((size_t)lrows * g.vq_n_embed);
        float* zq = c.arena.get<float>((size_t)lrows * g.vq_embed_dim);
        float* zz = c.arena.get<float>((size_t)lrows);

        launch_nchw_to_nhwc_pad(x_nchw + (long)i0 * g.vq_in_channels * RH * RW, ws.t, n, RH * RW, g.vq_in_channels, c.enc_cin_pad, s);
        Act x{ws.t, n, RH, RW, c.enc_cin_pad};
        conv3(x, c.enc_conv_in, ws.a, nullptr, 0, s);
        x = Act{ws.a, n, RH, RW, c.enc_conv_in.cout};
        auto pick2 = [&](float*& scratch, float*& y) {
            float* bufs[3] = {ws.a, ws.b, o};
            scratch = nullptr; y = nullptr;
            for (float* b : bufs) { if (b != x.p) { if (!scratch) scratch = b; else y = b; } }
        };
        auto res_step = [&](const ResBlockW& r) { float *sc, *y; pick2(sc, y); resblock(r, x, y, sc, ws, s, planes); };
        auto attn_step = [&](const AttnBlockW& ab) { float *sc, *y; pick2(sc, y); attnblock(ab, x, y, ws, s); };
        for (int lvl = 0; lvl < g.vq_num_levels; ++lvl) {
            const DownLevelW& d = c.down[lvl];
            for (size_t b = 0; b < d.blocks.size(); ++b) {
                res_step(d.blocks[b]);
                if (!d.attns.empty()) attn_step(d.attns[b]);
            }
            if (d.has_down) {  // F.pad(x, (0,1,0,1)) + conv 3x3 stride 2 padding 0
                float *sc, *y; pick2(sc, y);
                GemmArgs ga;
                ga.mode = MODE_CONV3;
                ga.A = x.p; ga.B = d.down.w; ga.C = y; ga.bias_n = d.down.b;
                ga.M = x.n * (x.h / 2) * (x.w / 2); ga.N = d.down.cout; ga.K = 9 * d.down.cin;
                ga.lda = d.down.cin; ga.ldb = 9 * d.down.cin; ga.ldc = d.down.cout;
                ga.conv_h = x.h / 2; ga.conv_w = x.w / 2; ga.conv_cin = d.down.cin; ga.conv_up = 0;
                ga.conv_hin = x.h; ga.conv_win = x.w; ga.conv_stride = 2; ga.conv_pad = 0;
                launch_gemm(ga, s);
                x = Act{y, x.n, x.h / 2, x.w / 2, d.down.cout};
            }
        }
        res_step(c.enc_mid1);
        attn_step(c.enc_mid_attn);
        res_step(c.enc_mid2);
        gn(x, c.enc_norm_out_w, c.enc_norm_out_b, ws.t, 1, ws, s, planes);
        Act t{ws.t, x.n, x.h, x.w, x.c};
        float *sc, *y; pick2(sc, y);
        conv3(t, c.enc_conv_out, y, nullptr, 0, s, planes);                     // [n, lat*lat, z_channels]
        conv1(y, lrows, c.quant_conv, zq, nullptr, s);                  // quant_conv (1x1)
        // distances to the codebook: (|z|^2 + |e|^2) - 2 z.e, arg-min (exact fp32 products: the codebook is never split)
        launch_row_sqnorm(zq, zz, lrows, g.vq_embed_dim, s);
        GemmArgs gd;
        gd.A = zq; gd.B = c.codebook; gd.C = dots;
        gd.M = (int)lrows; gd.N = g.vq_n_embed; gd.K = g.vq_embed_dim; gd.lda = g.vq_embed_dim; gd.ldb = g.vq_embed_dim; gd.ldc = g.vq_n_embed;
        launch_gemm(gd, s);
        launch_vq_argmin(dots, zz, c.codebook_sqnorm, ids + (long)i0 * lat_hw, lrows, g.vq_n_embed, s);
    }
}

}  // namespace bevgen
