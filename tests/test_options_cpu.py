"""How the library's modes reach the drop-in modules (bevgen_amd/modules/options.py): constructor keys next to the `_target_` (every reference constructor on the
path ends in **kwargs: gpt:270, ar_lm:55, muse_lm:56), $BEVGEN_* variables, defaults; no GPU."""
import pytest

from bevgen_amd import presets
from bevgen_amd.modules import options as O


def test_resolve_precedence(monkeypatch):
    for v in O.ENV.values():
        monkeypatch.delenv(v, raising=False)
    assert O.resolve(None, "ar") == {"precision": "f16x3", "weights": "f32", "kv_cache": "f32", "decode_weights": "f32", "decode_path": "auto"}
    assert set(O.resolve(None, "maskgit")) == {"precision", "weights"}
    monkeypatch.setenv("BEVGEN_KV_CACHE", "f16")
    monkeypatch.setenv("BEVGEN_DECODE_WEIGHTS", "f16")
    monkeypatch.setenv("BEVGEN_PRECISION", "fp32")
    r = O.resolve({"kv_cache": "f32"}, "ar")
    assert r["kv_cache"] == "f32" and r["decode_weights"] == "f16" and r["precision"] == "fp32"
    monkeypatch.setenv("BEVGEN_DECODE_PATH", "sideways")
    with pytest.raises(ValueError, match="decode_path"):
        O.resolve(None, "ar")
    with pytest.raises(ValueError, match="kv_cache"):
        O.pop_runtime_options({"kv_cache": "bf16"})


def test_gpt_and_net2net_take_modes_as_constructor_keys(monkeypatch):
    for v in O.ENV.values():
        monkeypatch.delenv(v, raising=False)
    from bevgen_amd.modules.stage1.vqgan import VQModel
    from bevgen_amd.modules.stage2.cond_transformer_multi_view import Net2NetTransformer
    from bevgen_amd.modules.transformer.mingpt_sparse import GPT

    cfg = presets.tiny_route_a(3, block=4)
    gpt = GPT(cfg, kv_cache="f16", some_reference_kwarg=1)
    assert gpt.runtime_options("ar")["kv_cache"] == "f16" and gpt.runtime_options("ar")["decode_weights"] == "f32"
    vq = VQModel(ddconfig=presets.VQ_DDCONFIG_TINY, n_embed=64, embed_dim=64)
    model = Net2NetTransformer(gpt, vq, None, kv_cache="f32", decode_weights="f16", precision="fp32", lr=1e-4)
    # the transformer's own key wins over the owner's, the rest is inherited; stage 1 inherits precision / weights only
    assert gpt.runtime_options("ar") == {"precision": "fp32", "weights": "f32", "kv_cache": "f16", "decode_weights": "f16", "decode_path": "auto"}
    assert vq.runtime_options("vq") == {"precision": "fp32", "weights": "f32"}
    assert model.lr == 1e-4 and not hasattr(model, "kv_cache")
    model.set_runtime_options(kv_cache="f32")          # explicit later change wins
    assert gpt.runtime_options("ar")["kv_cache"] == "f32"
    with pytest.raises(NotImplementedError, match="downsample_cond_size"):
        Net2NetTransformer(gpt, None, None, downsample_cond_size=8)


def test_maskgit_takes_modes(monkeypatch):
    for v in O.ENV.values():
        monkeypatch.delenv(v, raising=False)
    from bevgen_amd.modules.stage1.vqgan import VQModel
    from bevgen_amd.modules.stage2.cond_transformer_multi_view_muse import Net2NetTransformer
    from bevgen_amd.modules.stage2.muse_maskgit_pytorch import MaskGit, MaskGitTransformerMultiView

    cfg = presets.tiny_route_m(3, legacy=False, latent=(8, 8))
    tr = MaskGitTransformerMultiView(num_tokens=cfg.vocab_size, dim=cfg.num_embed, seq_len=cfg.cam_latent_res, depth=cfg.num_layers, dim_head=64,
                                     heads=cfg.num_heads, ff_mult=4, cfg=cfg, weights="f16")
    mg = MaskGit(image_size=cfg.cam_latent_res, transformer=tr, self_token_critic=True)
    assert mg.runtime_options("maskgit") == {"precision": "f16x3", "weights": "f16"}
    monkeypatch.setenv("BEVGEN_PRECISION", "fp32")
    assert mg.runtime_options("maskgit")["precision"] == "fp32"
    vq = VQModel(ddconfig=presets.VQ_DDCONFIG_TINY, n_embed=64, embed_dim=64)
    model = Net2NetTransformer(mg, vq, None, cfg, precision="f16x3")
    assert mg.runtime_options("maskgit") == {"precision": "f16x3", "weights": "f16"} and vq.runtime_options("vq")["precision"] == "f16x3"
    with pytest.raises(NotImplementedError, match="downsample_cond_size"):
        Net2NetTransformer(mg, vq, None, cfg, downsample_cond_size=16)
