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
}


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
