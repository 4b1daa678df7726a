"""Named parity cases shared by oracle/ref_import/make_golden.py and tests/ (TEST INFRASTRUCTURE).

A case fixes: configuration (bevgen_amd.presets), weight seed, input seed, batch size.  Weights are never stored:
they are regenerated from (seed, parameter name) by bevgen_amd.weights, so fixtures stay small.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Dict

import torch

from bevgen_amd import presets, synthetic, tables
from bevgen_amd import weights as W


@dataclass
class Case:
    name: str
    route: str  # 'm' | 'a'
    make_cfg: Callable
    weight_seed: int
    input_seed: int
    batch: int
    timesteps: int = 6  # Route M
    steps: int = 0  # Route A decode steps (0 = all)
    layout_seed: int = 0  # density < 1: torch seed under which the imported reference drew the per-layer layouts of the golden


CASES: Dict[str, Case] = {
    # tiny models (D=128, 2 heads, 2 layers): full tensors are stored
    "m_tiny_legacy": Case("m_tiny_legacy", "m", lambda: presets.tiny_route_m(3, legacy=True), 1234, 1, 2),
    "m_tiny_rays": Case("m_tiny_rays", "m", lambda: presets.tiny_route_m(3, legacy=False), 1234, 2, 2),
    "m_tiny_6cam": Case("m_tiny_6cam", "m", lambda: presets.tiny_route_m(6, legacy=False), 1234, 3, 1),
    "a_tiny_blk16": Case("a_tiny_blk16", "a", lambda: presets.tiny_route_a(3, block=16), 1234, 4, 2),
    "a_tiny_blk4": Case("a_tiny_blk4", "a", lambda: presets.tiny_route_a(3, block=4), 1234, 5, 3),
    "a_tiny_6cam": Case("a_tiny_6cam", "a", lambda: presets.tiny_route_a(6, block=16), 1234, 6, 1),
    # full-size models: token ids + a few logits rows only
    "a_config1": Case("a_config1", "a", presets.config1, 1234, 0, 1),  # BASELINE config 1: 24 layers, L=512, greedy, B=1
    "m_full_3cam": Case("m_full_3cam", "m", lambda: presets.config2(3), 1234, 0, 1, timesteps=18),  # released Argoverse shape
    # BASELINE config 2 (the bench workload: 6 views of 256x256, N=1536, L=1792) at full size, 4 MaskGit iterations
    "m_full_6cam": Case("m_full_6cam", "m", lambda: presets.config2(6), 1234, 0, 1, timesteps=4),
    # BASELINE config 4 at full size (nuScenes 6-view 224x400, L=2368, blk 16, centre-outward decode order): the first decode steps only
    "a_config4_head": Case("a_config4_head", "a", presets.config4, 1234, 0, 1, steps=6),
    # density < 1: every attention layer of the reference draws its own random per-head block layouts at construction (gpt:176, maskgen:217-228,
    # perm:125-143) and keeps them as the `master_layout` buffer; the goldens carry the layouts the imported reference drew (layout_seed)
    "a_tiny_d06": Case("a_tiny_d06", "a", lambda: presets.tiny_route_a(3, block=4, density=0.6), 1234, 7, 2, layout_seed=4242),
    "a_config4_d035_head": Case("a_config4_d035_head", "a", lambda: presets.config4(density=0.35), 1234, 0, 1, steps=6, layout_seed=4242),
}


def with_layer_layouts(sd, layouts):
    """state_dict with the per-layer layout buffers replaced by ``layouts`` [layers, H, L/blk, L/blk] (reference key names)."""
    out = dict(sd)
    for i in range(len(layouts)):
        out[f"blocks.{i}.attention.sparse_self_attention.master_layout"] = torch.as_tensor(layouts[i]).to(torch.int64)
    return out


def muse_kwargs(cfg):
    return dict(depth=cfg.num_layers, heads=cfg.num_heads, dim_head=64, ff_mult=4, num_tokens=cfg.vocab_size)


maskgit_state_dict = W.maskgit_state_dict
gpt_state_dict = W.gpt_state_dict
vq_state_dict = W.vq_state_dict


def inputs(case: Case, cfg):
    return synthetic.make_batch(cfg, case.batch, seed=case.input_seed)


def maskgit_noise(case: Case, cfg, seed: int = 5):
    rows = case.batch * cfg.num_cams
    T, V, ts = cfg.num_cam_tokens, cfg.vocab_size, case.timesteps
    return {"gumbel_u": synthetic.uniform_noise((ts, rows, T, V), seed, 0), "critic_u": synthetic.uniform_noise((ts, rows, T), seed, 1)}


VQ_TINY = dict(dd=presets.VQ_DDCONFIG_TINY, n_embed=64, embed_dim=64, seed=99, n_images=3)
# the released first-stage decoder at full size (f16: ch 128, ch_mult [1,1,2,2,4], 256x256, codebook 1024 x 256, configs/model/stage_2.yaml:36-55)
VQ_FULL = dict(dd=presets.VQ_DDCONFIG_F16, n_embed=1024, embed_dim=256, seed=99, n_images=1)
VQ_TINY_SEG = dict(dd=dict(presets.VQ_DDCONFIG_TINY, in_channels=7, out_ch=7), n_embed=64, embed_dim=64, seed=77, n_images=2)  # BEV cond stage (7 Argoverse classes)


def golden_state_dict(case: Case, cfg, g):
    """Weights of a Route A case; when the golden carries the per-layer layouts the imported reference drew (density < 1), they replace the
    configuration's layout in the `master_layout` buffers."""
    import numpy as np

    sd = gpt_state_dict(cfg, case.weight_seed)
    if "layer_layout_bits" in g.files:
        shape = tuple(int(v) for v in g["layer_layout_shape"])
        lay = np.unpackbits(g["layer_layout_bits"])[: int(np.prod(shape))].reshape(shape).astype(np.int64)
        sd = with_layer_layouts(sd, torch.from_numpy(lay))
    return sd
