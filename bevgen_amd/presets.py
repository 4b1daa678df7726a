"""Named configurations of the stage-2 sampling path (BASELINE.json ``configs`` and reduced test sizes).

Hyper-parameters come from the reference's YAML tree:
* Route A: configs/model/stage_2.yaml:6-34 (24 layers, blk 16, window 32, density 1.0)
* Route M: configs/experiment/muse_stage_two_multi_view.yaml:40-68 (14 layers, blk 1, camera bias + BEV embed, non-legacy prior)
* stage 1: configs/model/stage_2.yaml:36-58 (f16 VQGAN, ch 128, ch_mult [1,1,2,2,4], attn at 16)
"""
from __future__ import annotations

from typing import Any, Dict

from .config import GPTConfig
from . import synthetic

VQ_DDCONFIG_F16 = dict(double_z=False, z_channels=256, resolution=256, in_channels=3, out_ch=3, ch=128,
                       ch_mult=[1, 1, 2, 2, 4], num_res_blocks=2, attn_resolutions=[16], dropout=0.0)
VQ_DDCONFIG_F8_128 = dict(VQ_DDCONFIG_F16, resolution=128, ch_mult=[1, 1, 2, 4])  # config 1: 128x128 -> 16x16 latents
VQ_DDCONFIG_TINY = dict(double_z=False, z_channels=64, resolution=64, in_channels=3, out_ch=3, ch=32,
                        ch_mult=[1, 1, 2, 4], num_res_blocks=2, attn_resolutions=[8], dropout=0.0)  # 64x64 -> 8x8 latents

_COMMON = dict(embd_pdrop=0.0, resid_pdrop=0.0, attn_pdrop=0.0, n_unmasked=0, plot=False, backend="hip", image_embed=True)


def _cfg(**kw) -> GPTConfig:
    d = dict(_COMMON)
    d.update(kw)
    return GPTConfig(**d)


def _with_calibration(kw: Dict[str, Any]) -> Dict[str, Any]:
    if not kw.get("legacy_prob_matrix", True):
        n = 6 if kw["dataset"] == "NUSCENES" else kw["num_cams"]
        intr, extr = synthetic.rig_calibration(n)
        kw = dict(kw, cam_intrinsics=intr, cam_extrinsics=extr)
    return kw


def route_m(num_cams: int = 6, *, num_layers: int = 14, dim: int = 1024, heads: int = 16, vocab: int = 1024, cam_res=(256, 256),
            cam_latent_res=(16, 16), bev_latent_res=(16, 16), legacy_prob_matrix: bool = False) -> GPTConfig:
    """MaskGit route.  num_cams 3 = released Argoverse rig, 6 = nuScenes camera naming at 256x256 (BASELINE config 2)."""
    if num_cams == 3:
        rig, ds = "ARGOVERSE_FRONT_CAMERAS", "ARGOVERSE"
    elif num_cams == 6:
        rig, ds = "NUSCENES_CAMERAS", "NUSCENES"
    elif num_cams == 1:
        rig, ds = "NUSCENES_FRONT", "ARGOVERSE"
    else:
        raise ValueError(num_cams)
    kw = dict(num_layers=num_layers, num_heads=heads, num_embed=dim, hidden_size=dim, vocab_size=vocab, cond_vocab_size=vocab,
              num_cams=num_cams, window_len=32, density=1.0, sparse_block_size=1, cam_res=cam_res, cam_latent_res=cam_latent_res,
              bev_latent_res=bev_latent_res, camera_bias=True, bev_embed=True, cam_names=rig, dataset=ds, causal_order=True,
              legacy_prob_matrix=legacy_prob_matrix)
    return _cfg(**_with_calibration(kw))


def route_a(num_cams: int = 6, *, num_layers: int = 24, dim: int = 1024, heads: int = 16, vocab: int = 1024, cam_res=(224, 400),
            cam_latent_res=(14, 25), bev_latent_res=(16, 16), block: int = 16, window_len: int = 32, density: float = 1.0,
            camera_bias: bool = True, bev_embed: bool = True, dataset: str = None, rig: str = None) -> GPTConfig:
    """Autoregressive sparse-causal route.  Defaults = BASELINE config 4 (nuScenes 6-view 224x400, T=350, L=2368)."""
    if rig is None:
        rig, ds = {1: ("NUSCENES_FRONT", "ARGOVERSE"), 3: ("NUSCENES_ABLATION_CAMERAS", "NUSCENES"), 6: ("NUSCENES_CAMERAS", "NUSCENES")}[num_cams]
        dataset = dataset or ds
    kw = dict(num_layers=num_layers, num_heads=heads, num_embed=dim, hidden_size=dim, vocab_size=vocab, cond_vocab_size=vocab,
              num_cams=num_cams, window_len=window_len, density=density, sparse_block_size=block, cam_res=cam_res,
              cam_latent_res=cam_latent_res, bev_latent_res=bev_latent_res, camera_bias=camera_bias, bev_embed=bev_embed,
              cam_names=rig, dataset=dataset, causal_order=True, legacy_prob_matrix=True)
    return _cfg(**kw)


def config1() -> GPTConfig:
    """BASELINE config 1: single camera 128x128, 16x16 latents, K=256, L=512, Route A 24 layers, greedy, B=1."""
    return route_a(1, cam_res=(128, 128), cam_latent_res=(16, 16))


def config2(num_cams: int = 6) -> GPTConfig:
    """BASELINE config 2/3: Route M released hyper-parameters, 6 views of 256x256 (N=1536, L=1792)."""
    return route_m(num_cams)


def config4(density: float = 1.0) -> GPTConfig:
    """BASELINE config 4: Route A, nuScenes 6-view 224x400 (N=2100, L=2368, pad 12); density < 1 = the random per-head layout variant."""
    return route_a(6, density=density)


def tiny_route_m(num_cams: int = 3, legacy: bool = True, latent=(4, 4), bev=(4, 4)) -> GPTConfig:
    return route_m(num_cams, num_layers=2, dim=128, heads=2, vocab=64, cam_res=(64, 64), cam_latent_res=latent,
                   bev_latent_res=bev, legacy_prob_matrix=legacy)


def tiny_route_a(num_cams: int = 3, block: int = 16, density: float = 1.0) -> GPTConfig:
    return route_a(num_cams, num_layers=2, dim=128, heads=2, vocab=64, cam_res=(64, 64), cam_latent_res=(4, 5),
                   bev_latent_res=(4, 4), block=block, window_len=8, density=density)
