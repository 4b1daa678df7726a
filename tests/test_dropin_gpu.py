"""Drop-in classes on the GPU: the reference-shaped API (Net2NetTransformer.forward(batch) -> {'gen','rec','gt'}) runs on libbevgen_hip and
matches the oracle."""
import numpy as np
import pytest
import torch

from bevgen_amd import presets, synthetic, weights as W
from oracle import restate as R

pytestmark = pytest.mark.gpu


def test_net2net_route_m_forward_batch():
    from bevgen_amd.modules.stage1.vqgan import VQModel
    from bevgen_amd.modules.stage2.cond_transformer_multi_view_muse import Net2NetTransformer
    from bevgen_amd.modules.stage2.muse_maskgit_pytorch import MaskGit, MaskGitTransformerMultiView

    cfg = presets.tiny_route_m(3, legacy=False, latent=(8, 8))
    dd = presets.VQ_DDCONFIG_TINY
    tr = MaskGitTransformerMultiView(num_tokens=cfg.vocab_size, dim=cfg.num_embed, seq_len=cfg.cam_latent_res, depth=cfg.num_layers, dim_head=64,
                                     heads=cfg.num_heads, ff_mult=4, cfg=cfg)
    mg = MaskGit(image_size=cfg.cam_latent_res, transformer=tr, self_token_critic=True, cond_drop_prob=0.1)
    vq = VQModel(ddconfig=dd, n_embed=64, embed_dim=64, cam_res=(64, 64), cam_latent_res=(8, 8), cam_emd_dim=64)
    model = Net2NetTransformer(mg, vq, None, cfg, sample_iterations=5)
    sd_m = W.maskgit_state_dict(cfg, 1234)
    sd_v = W.vq_state_dict(dd, 64, 64, 99, with_encoder=True)
    full = {("maskgit." + k): v for k, v in sd_m.items()}
    full.update({("first_stage_model." + k): v for k, v in sd_v.items()})
    missing, unexpected = model.load_state_dict(full, strict=False)
    assert not unexpected and all(k.startswith("cond_stage") for k in missing)
    model = model.to("cuda")
    B = 2
    bt = synthetic.make_batch(cfg, B, seed=4)
    g = torch.Generator().manual_seed(0)
    images = torch.randn(B, cfg.num_cams, 64, 64, 3, generator=g)
    batch = {"cond_ids": bt["cond_ids"], "intrinsics_inv": bt["intrinsics_inv"], "extrinsics_inv": bt["extrinsics_inv"], "image": images}
    out = model.log_images(batch, noise="greedy")
    assert set(out) == {"gen", "rec", "gt"}
    gen = out["gen"].cpu()
    assert gen.shape == (B, cfg.num_cams, 3, 64, 64) and gen.min() >= 0 and gen.max() <= 1
    ids = R.maskgit_generate(sd_m, cfg, bt["cond_ids"], bt["intrinsics_inv"], bt["extrinsics_inv"], depth=cfg.num_layers, heads=cfg.num_heads, timesteps=5)
    ref = R.vq_decode_ids(sd_v, dd, ids.reshape(B * cfg.num_cams, -1), (8, 8), denorm=True).reshape(gen.shape)
    assert (gen - ref).abs().max() < 1e-3
    gt_ref = R.denormalize(images.movedim(-1, -3).reshape(-1, 3, 64, 64)).reshape(B, cfg.num_cams, 3, 64, 64)
    assert (out["gt"].cpu() - gt_ref).abs().max() < 1e-6
    # stochastic default path runs and stays in range
    out2 = model(batch)
    assert out2["gen"].shape == gen.shape


def test_gpt_forward_and_ar_net2net_sample():
    from bevgen_amd.modules.stage2.cond_transformer_multi_view import Net2NetTransformer
    from bevgen_amd.modules.transformer.mingpt_sparse import GPT

    cfg = presets.tiny_route_a(3, block=4)
    sd = W.gpt_state_dict(cfg, 1234)
    gpt = GPT(cfg)
    gpt.load_state_dict(sd)
    gpt = gpt.to("cuda")
    B = 2
    bt = synthetic.make_batch(cfg, B, seed=2)
    batch = {"intrinsics_inv": bt["intrinsics_inv"].cuda(), "extrinsics_inv": bt["extrinsics_inv"].cuda()}
    g = torch.Generator().manual_seed(1)
    ids = torch.randint(0, cfg.vocab_size, (B, cfg.num_cams, cfg.num_cam_tokens), generator=g)
    logits = gpt(ids.cuda(), bt["cond_ids"].cuda(), batch, sampling=True).cpu()
    ref = R.gpt_forward(sd, cfg, ids, bt["cond_ids"], bt["intrinsics_inv"], bt["extrinsics_inv"])
    assert ((logits - ref).abs().max() / ref.abs().max()) < 1e-4
    model = Net2NetTransformer(gpt, None, None)
    x = model.sample(None, bt["cond_ids"].cuda(), batch, sample=False).cpu()
    assert torch.equal(x, R.ar_sample_cached(sd, cfg, bt["cond_ids"], bt["intrinsics_inv"], bt["extrinsics_inv"]))
    assert torch.isinf(model.top_k_logits(torch.tensor([[1.0, 3.0, 3.0, 2.0, 0.0]]), 2)).sum() == 3


def test_sparse_self_attention_module_fp16_like_deepspeed():
    from bevgen_amd.modules.transformer.sparse_self_attention import CustomSparsityConfig, SparseSelfAttention

    cfg = presets.tiny_route_a(3, block=16)
    L, H = cfg.gpt_block_size, cfg.num_heads
    attn = SparseSelfAttention(CustomSparsityConfig(num_heads=H, layout=cfg.layout, block=16), attn_mask_mode="mul").cuda()
    g = torch.Generator().manual_seed(2)
    q, k, v = (torch.randn(2, H, L, 64, generator=g).half() for _ in range(3))
    add = torch.randn(1, L, L, generator=g)
    out = attn(q.cuda(), k.cuda(), v.cuda(), attn_mask=cfg.attention_mask.cuda(), add_mask=add.cuda())
    assert out.dtype == torch.float16
    ref = R.sparse_self_attention_dense(q.float(), k.float(), v.float(), cfg.layout, 16, cfg.attention_mask, add)
    assert (out.float().cpu() - ref).abs().max() < 2e-3  # fp16 output rounding
    with pytest.raises(ValueError, match="dividable"):
        attn.get_layout(L + 1)


def test_net2net_fully_native_forward_from_raw_batch():
    """forward(batch) with the reference's raw batch keys only (image, segmentation, camera matrices): BEV segmentation -> HIP encoder -> cond ids ->
    MaskGit -> HIP decoder, plus 'rec' through encode_to_z/decode and 'gt' - every stage checked against the oracle."""
    from bevgen_amd.modules.stage1.vqgan import VQModel, VQSegmentationModel
    from bevgen_amd.modules.stage2.cond_transformer_multi_view_muse import Net2NetTransformer
    from bevgen_amd.modules.stage2.muse_maskgit_pytorch import MaskGit, MaskGitTransformerMultiView
    from oracle import cases

    cfg = presets.tiny_route_m(3, legacy=False, latent=(8, 8), bev=(8, 8))
    dd, dds = cases.VQ_TINY["dd"], cases.VQ_TINY_SEG["dd"]
    tr = MaskGitTransformerMultiView(num_tokens=cfg.vocab_size, dim=cfg.num_embed, seq_len=cfg.cam_latent_res, depth=cfg.num_layers, dim_head=64,
                                     heads=cfg.num_heads, ff_mult=4, cfg=cfg)
    mg = MaskGit(image_size=cfg.cam_latent_res, transformer=tr, self_token_critic=True, cond_drop_prob=0.1)
    vq = VQModel(ddconfig=dd, n_embed=64, embed_dim=64, cam_res=(64, 64), cam_latent_res=(8, 8), cam_emd_dim=64)
    vqc = VQSegmentationModel(n_labels=7, ddconfig=dds, n_embed=64, embed_dim=64, cam_res=(64, 64), cam_latent_res=(8, 8), cam_emd_dim=64, image_key="segmentation")
    model = Net2NetTransformer(mg, vq, vqc, cfg, sample_iterations=4)
    sd_m = W.maskgit_state_dict(cfg, 1234)
    sd_v = W.vq_state_dict(dd, 64, 64, 99, with_encoder=True)
    sd_c = W.vq_state_dict(dds, 64, 64, 77, with_encoder=True)
    full = {("maskgit." + k): v for k, v in sd_m.items()}
    full.update({("first_stage_model." + k): v for k, v in sd_v.items()})
    full.update({("cond_stage_model." + k): v for k, v in sd_c.items()})
    missing, unexpected = model.load_state_dict(full, strict=False)
    assert not unexpected and missing == ["cond_stage_model.colorize"], (missing, unexpected)
    model = model.to("cuda")
    B = 2
    bt = synthetic.make_batch(cfg, B, seed=8)
    g = torch.Generator().manual_seed(5)
    batch = {"image": torch.randn(B, cfg.num_cams, 64, 64, 3, generator=g), "segmentation": torch.randn(B, 64, 64, 7, generator=g),
             "intrinsics_inv": bt["intrinsics_inv"], "extrinsics_inv": bt["extrinsics_inv"]}
    out = model.log_images(batch, noise="greedy")
    # oracle chain
    seg = batch["segmentation"].movedim(-1, -3)
    c_ids = R.vq_encode_ids(sd_c, dds, seg)
    ids = R.maskgit_generate(sd_m, cfg, c_ids, bt["intrinsics_inv"], bt["extrinsics_inv"], depth=cfg.num_layers, heads=cfg.num_heads, timesteps=4)
    gen_ref = R.vq_decode_ids(sd_v, dd, ids.reshape(B * cfg.num_cams, -1), (8, 8), denorm=True).reshape(B, cfg.num_cams, 3, 64, 64)
    img = batch["image"].movedim(-1, -3).reshape(B * cfg.num_cams, 3, 64, 64)
    z_ids = R.vq_encode_ids(sd_v, dd, img)
    rec_ref = R.vq_decode_ids(sd_v, dd, z_ids, (8, 8), denorm=True).reshape(B, cfg.num_cams, 3, 64, 64)
    assert (out["gen"].cpu() - gen_ref).abs().max() < 1e-3
    assert (out["rec"].cpu() - rec_ref).abs().max() < 1e-3
    assert (out["gt"].cpu() - R.denormalize(img).reshape(B, cfg.num_cams, 3, 64, 64)).abs().max() < 1e-6


def test_route_a_net2net_forward_from_raw_batch_with_partial_decoding():
    """Route A drop-in (ar_lm Net2NetTransformer): forward / log_images with the reference's raw batch keys only - BEV segmentation -> HIP encoder -> cond ids,
    images -> HIP encoder -> z ids for 'rec' and for partial decoding (partial_decoding=4 fixes cameras [3, 0, 2] deterministically, ar_lm:509-510),
    greedy sampling through prefill + KV-cache decode, non-square 4 x 5 latents through the fully convolutional decoder - every stage against the oracle."""
    from bevgen_amd.modules.stage1.vqgan import VQModel, VQSegmentationModel
    from bevgen_amd.modules.stage2.cond_transformer_multi_view import Net2NetTransformer
    from bevgen_amd.modules.transformer.mingpt_sparse import GPT
    from oracle import cases

    cfg = presets.route_a(6, num_layers=2, dim=128, heads=2, vocab=64, cam_res=(32, 40), cam_latent_res=(4, 5), bev_latent_res=(8, 8), block=16, window_len=8)
    dd, dds = cases.VQ_TINY["dd"], cases.VQ_TINY_SEG["dd"]
    gpt = GPT(cfg)
    vq = VQModel(ddconfig=dd, n_embed=64, embed_dim=64, cam_res=(32, 40), cam_latent_res=(4, 5), cam_emd_dim=64)
    vqc = VQSegmentationModel(n_labels=7, ddconfig=dds, n_embed=64, embed_dim=64, cam_res=(64, 64), cam_latent_res=(8, 8), cam_emd_dim=64, image_key="segmentation")
    model = Net2NetTransformer(gpt, vq, vqc, partial_decoding=4)
    sd_g = W.gpt_state_dict(cfg, 1234)
    sd_v = W.vq_state_dict(dd, 64, 64, 99, with_encoder=True)
    sd_c = W.vq_state_dict(dds, 64, 64, 77, with_encoder=True)
    full = {("transformer." + k): v for k, v in sd_g.items()}
    full.update({("first_stage_model." + k): v for k, v in sd_v.items()})
    full.update({("cond_stage_model." + k): v for k, v in sd_c.items()})
    missing, unexpected = model.load_state_dict(full, strict=False)
    assert not unexpected and missing == ["cond_stage_model.colorize"], (missing, unexpected)
    model = model.to("cuda")
    B, C, T = 2, cfg.num_cams, cfg.num_cam_tokens
    bt = synthetic.make_batch(cfg, B, seed=9)
    g = torch.Generator().manual_seed(6)
    batch = {"image": torch.randn(B, C, 32, 40, 3, generator=g), "segmentation": torch.randn(B, 64, 64, 7, generator=g),
             "intrinsics_inv": bt["intrinsics_inv"], "extrinsics_inv": bt["extrinsics_inv"]}
    out = model.log_images(batch, sample=False)          # greedy instead of the reference's top-k-100 draw: comparable with the oracle
    # oracle chain
    c_ids = R.vq_encode_ids(sd_c, dds, batch["segmentation"].movedim(-1, -3))
    img = batch["image"].movedim(-1, -3).reshape(B * C, 3, 32, 40)
    z_ids = R.vq_encode_ids(sd_v, dd, img)
    forced = R.partial_forced_ids(cfg, [3, 0, 2], z_ids.reshape(B, C, T))
    x = R.ar_sample_cached(sd_g, cfg, c_ids, bt["intrinsics_inv"], bt["extrinsics_inv"], forced_ids=forced)
    assert torch.equal(x[:, [0, 2, 3]], z_ids.reshape(B, C, T)[:, [0, 2, 3]])
    gen_ref = R.vq_decode_ids(sd_v, dd, x.reshape(B * C, T), (4, 5), denorm=True).reshape(B, C, 3, 32, 40)
    rec_ref = R.vq_decode_ids(sd_v, dd, z_ids, (4, 5), denorm=True).reshape(B, C, 3, 32, 40)
    assert tuple(out["gen"].shape) == (B, C, 3, 32, 40)
    assert (out["gen"].cpu() - gen_ref).abs().max() < 1e-3
    assert (out["rec"].cpu() - rec_ref).abs().max() < 1e-3
    assert (out["gt"].cpu() - R.denormalize(img).reshape(B, C, 3, 32, 40)).abs().max() < 1e-6
    # forward() = log_images(generate_only=True) with the module's top_k: the stochastic default path runs and returns the three entries
    out2 = model(batch)
    assert set(out2) == {"gen", "rec", "gt"} and tuple(out2["gen"].shape) == (B, C, 3, 32, 40) and float(out2["gen"].min()) >= 0 and float(out2["gen"].max()) <= 1


@pytest.mark.parametrize("name", ["a_tiny_d06", "a_config4_d035_head"])
def test_gpt_dropin_keeps_every_layers_master_layout(name):
    """density < 1: the reference draws one random layout PER attention layer (gpt:176, maskgen:217-251) and keeps them in the checkpoint as
    blocks.{i}...master_layout.  The drop-in GPT must hand every one of them to the library (round 2 uploaded only block 0's): tokens of the module path
    = the tokens the imported reference generated with those layouts."""
    from conftest import golden
    from bevgen_amd.modules.stage2.cond_transformer_multi_view import Net2NetTransformer
    from bevgen_amd.modules.transformer.mingpt_sparse import GPT
    from oracle import cases

    case = cases.CASES[name]
    g = golden("route_a_" + name)
    cfg = case.make_cfg()
    sd = cases.golden_state_dict(case, cfg, g)
    lay = [sd[f"blocks.{i}.attention.sparse_self_attention.master_layout"] for i in range(cfg.num_layers)]
    assert any(not torch.equal(lay[0], l) for l in lay[1:]), "fixture must carry different layouts per layer"
    gpt = GPT(cfg)
    missing, unexpected = gpt.load_state_dict(sd)
    assert not missing and not unexpected
    gpt = gpt.to("cuda")
    cond = torch.from_numpy(g["cond_ids"]).long().cuda()
    batch = {"intrinsics_inv": torch.from_numpy(g["I_inv"]).cuda(), "extrinsics_inv": torch.from_numpy(g["E_inv"]).cuda()}
    want = torch.from_numpy(g["sample_greedy"]).long()
    if case.steps == 0:
        x = Net2NetTransformer(gpt, None, None).sample(None, cond, batch, sample=False).cpu()
    else:   # full-size model: the golden holds the first `steps` tokens only
        x = gpt.context().ar_sample(cond, batch["intrinsics_inv"], batch["extrinsics_inv"], greedy=True, steps=case.steps).cpu()
    assert torch.equal(x, want), f"{(x != want).sum().item()} tokens differ from the reference's (per-layer layouts lost?)"


def test_maskgit_dropin_generate_without_token_critic():
    """MaskGit.generate(force_not_use_token_critic=True[, can_remask_prev_masked=True]) and a MaskGit built without any critic (muse_net:553, 611-622) through the
    drop-in module: tokens = what the imported reference generated."""
    from conftest import golden
    from bevgen_amd.modules.stage2.muse_maskgit_pytorch import MaskGit, MaskGitTransformerMultiView
    from oracle import cases

    case = cases.CASES["m_tiny_rays"]
    g = golden("route_m_branches_m_tiny_rays")
    cfg = case.make_cfg()
    sd = cases.maskgit_state_dict(cfg, case.weight_seed)
    cond = torch.from_numpy(g["cond_ids"]).long().cuda()
    batch = {"intrinsics_inv": torch.from_numpy(g["I_inv"]).cuda(), "extrinsics_inv": torch.from_numpy(g["E_inv"]).cuda()}

    def build(self_critic, **kw):
        tr = MaskGitTransformerMultiView(num_tokens=cfg.vocab_size, dim=cfg.num_embed, seq_len=cfg.cam_latent_res, depth=cfg.num_layers, dim_head=64, heads=cfg.num_heads,
                                         ff_mult=4, cfg=cfg)
        mg = MaskGit(image_size=cfg.cam_latent_res, transformer=tr, self_token_critic=self_critic, **kw)
        want = {k: v for k, v in sd.items() if self_critic or not k.startswith("token_critic.")}
        missing, unexpected = mg.load_state_dict(want, strict=False)
        assert not missing and not unexpected, (missing, unexpected)
        return mg.to("cuda")

    mg = build(True, no_mask_token_prob=0.1)
    for tag, kw in (("nocritic", dict(force_not_use_token_critic=True)), ("nocritic_remask", dict(force_not_use_token_critic=True, can_remask_prev_masked=True))):
        x = mg.generate(cond_images=cond, fmap_size=cfg.cam_latent_res, batch=batch, timesteps=case.timesteps, noise="greedy", **kw).cpu()
        assert torch.equal(x, torch.from_numpy(g[f"gen_{tag}_greedy"]).long()), tag
    with pytest.raises(AssertionError, match="non-masked tokens"):
        build(True).generate(cond_images=cond, fmap_size=cfg.cam_latent_res, batch=batch, timesteps=case.timesteps, noise="greedy", force_not_use_token_critic=True,
                             can_remask_prev_masked=True)
    bare = build(False)   # no critic at all: use_token_critic is False without the flag (muse_net:553)
    x = bare.generate(cond_images=cond, fmap_size=cfg.cam_latent_res, batch=batch, timesteps=case.timesteps, noise="greedy").cpu()
    assert torch.equal(x, torch.from_numpy(g["gen_nocritic_greedy"]).long())


def test_net2net_sample_in_the_fp16_decode_mode_through_the_dropin_boundary(monkeypatch):
    """The benchmarked decode mode (fp16 KV cache + fp16 decode weights, fused layer) is reachable from the reference's plugin boundary: as constructor keys next to the
    `_target_` (GPT(cfg, **kwargs), gpt:270 / Net2NetTransformer(..., **kwargs), ar_lm:55) and by $BEVGEN_KV_CACHE / $BEVGEN_DECODE_WEIGHTS.  Tokens through
    Net2NetTransformer.sample (ar_lm:154-227) = the tokens of a Context built directly in that mode = the rounded-weights oracle's up to the first near-tie."""
    from bevgen_amd.modules import options as O
    from bevgen_amd.modules.stage2.cond_transformer_multi_view import Net2NetTransformer
    from bevgen_amd.modules.transformer.mingpt_sparse import GPT
    from bevgen_amd.runtime import Context

    for v in O.ENV.values():
        monkeypatch.delenv(v, raising=False)
    cfg = presets.route_a(3, num_layers=2, dim=256, heads=4, vocab=64, cam_res=(64, 64), cam_latent_res=(4, 5), bev_latent_res=(4, 4), block=16, window_len=8)
    sd = W.gpt_state_dict(cfg, 1234)
    B = 3
    bt = synthetic.make_batch(cfg, B, seed=3)
    batch = {"intrinsics_inv": bt["intrinsics_inv"].cuda(), "extrinsics_inv": bt["extrinsics_inv"].cuda()}
    direct = Context(cfg, route="ar", precision="f16x3", kv_cache="f16", decode_weights="f16", decode_path="fused")
    direct.load_state_dict(sd)
    direct.set_tables()
    direct.finalize()
    want = direct.ar_sample(bt["cond_ids"], bt["intrinsics_inv"], bt["extrinsics_inv"], greedy=True).cpu()
    direct.close()

    # (1) keys on the outer module, as `+model.kv_cache=f16 +model.decode_weights=f16 +model.decode_path=fused` would put them
    gpt = GPT(cfg)
    gpt.load_state_dict(sd)
    model = Net2NetTransformer(gpt.to("cuda"), None, None, kv_cache="f16", decode_weights="f16", decode_path="fused")
    x = model.sample(None, bt["cond_ids"].cuda(), batch, sample=False).cpu()
    c = gpt.context()
    assert (c.kv_cache, c.decode_weights, c.decode_path, c.precision) == ("f16", "f16", "fused", "f16x3")
    assert torch.equal(x, want)
    # (2) keys on the transformer's own `_target_` node
    gpt2 = GPT(cfg, kv_cache="f16", decode_weights="f16", decode_path="fused")
    gpt2.load_state_dict(sd)
    x2 = Net2NetTransformer(gpt2.to("cuda"), None, None).sample(None, bt["cond_ids"].cuda(), batch, sample=False).cpu()
    assert torch.equal(x2, want)
    # (3) process-wide by environment
    monkeypatch.setenv("BEVGEN_KV_CACHE", "f16")
    monkeypatch.setenv("BEVGEN_DECODE_WEIGHTS", "f16")
    monkeypatch.setenv("BEVGEN_DECODE_PATH", "fused")
    gpt3 = GPT(cfg)
    gpt3.load_state_dict(sd)
    x3 = Net2NetTransformer(gpt3.to("cuda"), None, None).sample(None, bt["cond_ids"].cuda(), batch, sample=False).cpu()
    assert gpt3.context().kv_cache == "f16" and torch.equal(x3, want)
    # and the mode is the rounded-weights model: with the fp32 cache the oracle on the rounded state_dict is reproduced token for token
    sdr = dict(sd)
    for k, v in sd.items():
        if k == "head.weight" or (k.startswith("blocks.") and k.endswith(".weight") and any(t in k for t in (".attention.query.", ".attention.key.", ".attention.value.", ".mlp.0.", ".mlp.2."))):
            sdr[k] = v.half().float()
    model.set_runtime_options(kv_cache="f32")
    xe = model.sample(None, bt["cond_ids"].cuda(), batch, sample=False).cpu()
    assert gpt.context().kv_cache == "f32"
    assert torch.equal(xe, R.ar_sample_cached(sdr, cfg, bt["cond_ids"], bt["intrinsics_inv"], bt["extrinsics_inv"]))


def test_dropin_default_precision_is_the_benchmarked_one(monkeypatch):
    """The drop-in modules default to the split-precision products (f16x3: the mode bench.py's headline runs in; token-exact on every fixture); the exact fp32 MFMA mode
    is `precision: fp32` / $BEVGEN_PRECISION=fp32."""
    from bevgen_amd.modules import options as O
    from bevgen_amd.modules.stage2.muse_maskgit_pytorch import MaskGit, MaskGitTransformerMultiView

    for v in O.ENV.values():
        monkeypatch.delenv(v, raising=False)
    cfg = presets.tiny_route_m(3, legacy=False, latent=(8, 8))
    tr = MaskGitTransformerMultiView(num_tokens=cfg.vocab_size, dim=cfg.num_embed, seq_len=cfg.cam_latent_res, depth=cfg.num_layers, dim_head=64,
                                     heads=cfg.num_heads, ff_mult=4, cfg=cfg)
    mg = MaskGit(image_size=cfg.cam_latent_res, transformer=tr, self_token_critic=True).to("cuda")
    assert mg.context().precision == "f16x3"
    mg.set_runtime_options(precision="fp32")
    assert mg.context().precision == "fp32"
