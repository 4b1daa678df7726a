"""End-to-end entry point on the GPU: reference-style config tree (`_target_`s naming the REFERENCE classes) -> hydra_lite -> drop-ins -> libbevgen_hip
-> GenerateImages on-disk layout (generate.py:26-77 + utils/callback.py:72-132)."""
import os
import textwrap

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

MAIN = """
# @package _global_
defaults:
  - _self_
  - model: tiny_muse
  - callbacks: default
  - modes: null
seed: 3
cam_res: [64, 64]
cam_latent_res: [8, 8]
num_cams: 3
datamodule:
  batch_size: 2
  test:
    eval_generate: ${oc.env:BEVGEN_TEST_OUT}
"""
MODEL = """
_target_: multi_view_generation.modules.stage2.cond_transformer_multi_view_muse.Net2NetTransformer
ckpt_path: does/not/exist.ckpt
sample_iterations: 4
transformer: null
maskgit:
  _target_: multi_view_generation.modules.stage2.muse_maskgit_pytorch.MaskGit
  image_size: ${cam_latent_res}
  cond_drop_prob: 0.1
  self_token_critic: True
  transformer:
    _target_: multi_view_generation.modules.stage2.muse_maskgit_pytorch.MaskGitTransformerMultiView
    num_tokens: ${model.first_stage.n_embed}
    seq_len: ${cam_latent_res}
    dim: ${model.cfg.num_embed}
    depth: ${model.cfg.num_layers}
    dim_head: 64
    heads: ${model.cfg.num_heads}
    ff_mult: 4
    cfg: ${model.cfg}
cfg:
  _target_: multi_view_generation.modules.transformer.mingpt_sparse.GPTConfig
  num_cams: ${num_cams}
  vocab_size: ${model.first_stage.n_embed}
  cond_vocab_size: ${model.cond_stage.n_embed}
  hidden_size: 128
  num_embed: 128
  num_heads: 2
  num_layers: 2
  backend: deepspeed
  sparse_block_size: 1
  window_len: 32
  cam_res: ${cam_res}
  cam_latent_res: ${cam_latent_res}
  causal_order: True
  camera_bias: True
  image_embed: True
  bev_embed: True
  bev_latent_res: [4, 4]
  density: 1.0
  cam_names: ARGOVERSE_FRONT_CAMERAS
  dataset: ARGOVERSE
  legacy_prob_matrix: true
first_stage:
  _target_: multi_view_generation.modules.stage1.vqgan.VQModel
  ckpt_path: does/not/exist_either.ckpt
  denormalize: True
  embed_dim: 64
  n_embed: 64
  cam_res: ${cam_res}
  cam_latent_res: ${cam_latent_res}
  cam_emd_dim: 64
  ddconfig: {double_z: False, z_channels: 64, resolution: 64, in_channels: 3, out_ch: 3, ch: 32, ch_mult: [1, 1, 2, 4], num_res_blocks: 1, attn_resolutions: [8], dropout: 0.0}
  lossconfig:
    _target_: multi_view_generation.modules.losses.vqperceptual.DummyLoss
cond_stage:
  _target_: multi_view_generation.modules.stage1.vqgan.VQSegmentationModel
  ckpt_path: null
  embed_dim: 64
  n_embed: 64
  image_key: segmentation
  n_labels: 7
  denormalize: False
  cam_res: ${cam_res}
  cam_latent_res: ${cam_latent_res}
  cam_emd_dim: 64
  ddconfig: {double_z: False, z_channels: 64, resolution: 64, in_channels: 7, out_ch: 7, ch: 32, ch_mult: [1, 1, 2, 4], num_res_blocks: 1, attn_resolutions: [8], dropout: 0.0}
  lossconfig:
    _target_: multi_view_generation.modules.losses.vqperceptual.DummyLoss
"""
CALLBACKS = """
image_logger:
  _target_: multi_view_generation.utils.GenerateImages
  save_dir: ${datamodule.test.eval_generate}
"""


def test_generate_entry_point_end_to_end(tmp_path, monkeypatch):
    from PIL import Image
    from bevgen_amd import generate

    cfgdir = tmp_path / "configs"
    for rel, text in {"train.yaml": MAIN, "model/tiny_muse.yaml": MODEL, "callbacks/default.yaml": CALLBACKS}.items():
        p = cfgdir / rel
        p.parent.mkdir(parents=True, exist_ok=True)
        p.write_text(textwrap.dedent(text).lstrip("\n"))
    out = tmp_path / "out"
    monkeypatch.setenv("BEVGEN_TEST_OUT", str(out))
    rc = generate.main(["--config-dir", str(cfgdir), "--synthetic", "3", "--random-weights", "model.cfg.bev_latent_res=[4,4]"])
    assert rc == 0
    toks = sorted(os.listdir(out / "sample"))
    assert toks == ["synthetic_000000", "synthetic_000001", "synthetic_000002"]
    cams = ("ring_front_left", "ring_front_center", "ring_front_right")
    for t in toks:
        assert sorted(os.listdir(out / "sample" / t)) == sorted([f"{c}.jpg" for c in cams] + ["bev.npz"])
        img = np.asarray(Image.open(out / "sample" / t / "ring_front_center.jpg"))
        assert img.shape == (64, 64, 3) and img.std() > 1.0          # a decoded picture, not a constant
        assert "arr_0" in np.load(out / "sample" / t / "bev.npz")
    # two scenes of one batch differ (different BEV condition ids)
    a = np.asarray(Image.open(out / "sample" / toks[0] / "ring_front_left.jpg")).astype(np.int32)
    b = np.asarray(Image.open(out / "sample" / toks[1] / "ring_front_left.jpg")).astype(np.int32)
    assert np.abs(a - b).mean() > 0.5
