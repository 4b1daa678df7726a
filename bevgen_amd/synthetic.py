"""Seeded synthetic inputs of the stage-2 sampling path (no datasets / checkpoints exist offline).

Shapes follow the collated ``batch`` dict the reference's dataset hands to ``Net2NetTransformer.forward``
(bev_utils/argoverse.py:296-305, bev_utils/util.py:50-71): ``intrinsics_inv [B,C,3,3]``, ``extrinsics_inv [B,C,4,4]``;
``cond_ids [B,K]`` are the BEV VQ token ids ``encode_to_c`` produces (muse_lm:149-155).
"""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np
import torch


def ring_cameras(B: int, C: int, cam_res: Tuple[int, int], seed: int = 0) -> Tuple[torch.Tensor, torch.Tensor]:
    """(I_inv [B,C,3,3], E_inv [B,C,4,4]): pinhole K = [[f,0,w/2],[0,f,h/2],[0,0,1]], f = 0.8*w; cameras on a yaw ring
    2*pi*i/C (+ small per-scene jitter) with translations U(-0.5,0.5) m; E maps ego -> camera, E_inv = E^-1."""
    g = np.random.Generator(np.random.Philox(key=[seed, 0xCA]))
    h, w = cam_res
    f = 0.8 * w
    Kmat = np.array([[f, 0, w / 2], [0, f, h / 2], [0, 0, 1]], dtype=np.float64)
    I_inv = np.broadcast_to(np.linalg.inv(Kmat), (B, C, 3, 3)).copy()
    E_inv = np.zeros((B, C, 4, 4), dtype=np.float64)
    axis = np.array([[0, 0, 1.0], [-1.0, 0, 0], [0, -1.0, 0]])  # camera (x right, y down, z fwd) -> ego (x fwd, y left, z up)
    for b in range(B):
        for i in range(C):
            yaw = 2 * np.pi * i / C + 0.05 * g.standard_normal()
            R = np.array([[np.cos(yaw), -np.sin(yaw), 0], [np.sin(yaw), np.cos(yaw), 0], [0, 0, 1.0]])
            cam_to_ego = np.eye(4)
            cam_to_ego[:3, :3] = R @ axis
            cam_to_ego[:3, 3] = g.uniform(-0.5, 0.5, size=3) + np.array([0, 0, 1.5])
            E_inv[b, i] = cam_to_ego  # inverse of the ego->camera extrinsic
    return torch.from_numpy(I_inv).float(), torch.from_numpy(E_inv).float()


def rig_calibration(n_cams: int, seed: int = 7) -> Tuple[torch.Tensor, torch.Tensor]:
    """([n,3,3] intrinsics, [n,4,4] extrinsics) standing in for pretrained/cam_data_<dataset>.pt (maskgen:89-98)."""
    g = np.random.Generator(np.random.Philox(key=[seed, 0xCB]))
    intr = np.zeros((n_cams, 3, 3))
    extr = np.zeros((n_cams, 4, 4))
    axis = np.array([[0, 0, 1.0], [-1.0, 0, 0], [0, -1.0, 0]])
    for i in range(n_cams):
        f = 1200.0 + 50 * i
        intr[i] = [[f, 0, 800.0], [0, f, 450.0], [0, 0, 1.0]]
        yaw = 2 * np.pi * i / n_cams
        R = np.array([[np.cos(yaw), -np.sin(yaw), 0], [np.sin(yaw), np.cos(yaw), 0], [0, 0, 1.0]])
        cam_to_ego = np.eye(4)
        cam_to_ego[:3, :3] = R @ axis
        cam_to_ego[:3, 3] = g.uniform(-0.5, 0.5, size=3)
        extr[i] = np.linalg.inv(cam_to_ego)
    return torch.from_numpy(intr).float(), torch.from_numpy(extr).float()


def bev_token_ids(B: int, K: int, vocab: int, seed: int = 0) -> torch.Tensor:
    g = np.random.Generator(np.random.Philox(key=[seed, 0xCC]))
    return torch.from_numpy(g.integers(0, vocab, size=(B, K), dtype=np.int64))


def uniform_noise(shape, seed: int, stream: int = 0) -> torch.Tensor:
    """Explicit U[0,1) noise (Philox) replacing the reference's in-place ``uniform_`` draws."""
    g = np.random.Generator(np.random.Philox(key=[seed, 0xD0 + stream]))
    return torch.from_numpy(g.random(size=shape, dtype=np.float32))


def make_batch(cfg, B: int, seed: int = 0) -> Dict[str, torch.Tensor]:
    I_inv, E_inv = ring_cameras(B, cfg.num_cams, cfg.cam_res, seed)
    return {
        "intrinsics_inv": I_inv,
        "extrinsics_inv": E_inv,
        "cond_ids": bev_token_ids(B, cfg.num_cond_tokens, cfg.cond_vocab_size, seed),
    }


def random_layer_layouts(cfg, seed: int = 4242):
    """Per-layer block layouts for density < 1 models, drawn the way the reference draws them at construction (one `multi_outward_pattern` call per attention
    layer, gpt:176 / maskgen:217-228, on torch's CPU generator): {state_dict key: int64 [H, L/blk, L/blk]} for every layer, and the fraction of the causal
    (decode row, key) pairs that sit in a present block (averaged over a sample of layers / heads) - the factor by which the algorithmic K/V bytes shrink."""
    import torch

    from . import tables

    state = torch.get_rng_state()
    torch.manual_seed(seed)
    pat = tables.attention_patterns(cfg)
    lays = [tables.head_layouts(cfg, pat) for _ in range(cfg.num_layers)]
    torch.set_rng_state(state)
    sd = {f"blocks.{i}.attention.sparse_self_attention.master_layout": lay.to(torch.int64) for i, lay in enumerate(lays)}
    blk, K, N = cfg.sparse_block_size, cfg.num_cond_tokens, cfg.num_img_tokens
    allowed = cfg.attention_mask != 0
    causal = torch.tril(torch.ones_like(allowed))
    rows = slice(K, K + N)
    tot = vis = 0.0
    for lay in lays[:4]:
        for h in range(0, cfg.num_heads, 4):
            full = lay[h].bool().repeat_interleave(blk, 0).repeat_interleave(blk, 1) & allowed
            tot += float((causal[rows] & allowed[rows]).sum())
            vis += float((causal[rows] & full[rows]).sum())
    return sd, vis / tot
