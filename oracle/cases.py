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
    heavy: int = 0  # != 0: trained-like heavy-tailed weights (heavy_tail below, this seed) instead of the reference's N(0, 0.02) initialisation


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
    # trained-like weights (heavy_tail): a few outlier channels x 30-100 in the LayerNorm gains and the MLP / feed-forward up-projection rows - activations
    # leave the O(1-10) range every N(0, 0.02) fixture lives in; the f16 hi / lo splits of precision = 'f16x3' must still give the reference's tokens
    "m_tiny_heavy": Case("m_tiny_heavy", "m", lambda: presets.tiny_route_m(3, legacy=False), 1234, 8, 2, heavy=31),
    "a_tiny_heavy": Case("a_tiny_heavy", "a", lambda: presets.tiny_route_a(3, block=16), 1234, 9, 2, heavy=37),
    # the smallest Route-M shapes on which the GEGLU epilogue and the folded LayerNorms of the HIP path engage (F = int(dim 8 / 3) a multiple of 64: dim 192 -> F 512),
    # with the reference initialisation and with heavy tails (the fold multiplies RAW residual rows by W o gamma and subtracts mean x column sums: cancellation is the risk)
    "m_fold": Case("m_fold", "m", lambda: presets.route_m(3, num_layers=2, dim=192, heads=3, vocab=64, cam_res=(64, 64), cam_latent_res=(4, 4), bev_latent_res=(4, 4)), 1234, 10, 2),
    "m_fold_heavy": Case("m_fold_heavy", "m", lambda: presets.route_m(3, num_layers=2, dim=192, heads=3, vocab=64, cam_res=(64, 64), cam_latent_res=(4, 4), bev_latent_res=(4, 4)), 1234, 11, 2, heavy=61),
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


def heavy_tail(sd, seed: int, n_out: int = 4, lo: float = 30.0, hi: float = 100.0):
    """Trained-like heavy tails on top of the reference-initialisation weights (TEST INFRASTRUCTURE): trained transformers carry a handful of outlier channels whose
    LayerNorm gains and up-projection rows are one to two orders of magnitude above the rest; the N(0, 0.02) initialisation (gpt:310-317) has none, so activations of every
    other fixture stay O(1-10).  For every tensor of the kinds below, ``n_out`` channels (drawn per tensor from ``seed``) are multiplied by a factor in [lo, hi]:
      * LayerNorm / GroupNorm gains:  ``*.ln1.weight, *.ln2.weight, ln_f.weight`` (Route A), ``*.norm.gamma, *.2.0.gamma, *.2.3.gamma`` (Route M), ``*.norm*.weight`` (VQGAN)
      * up-projection ROWS:           ``*.mlp.0.weight`` (+ its bias), ``*.2.1.weight`` (GEGLU up: x and gate halves), VQGAN ``nin_shortcut`` / ``conv_in`` output channels
    The aliased ``token_critic.net.*`` tensors of a MaskGit follow their ``transformer.*`` originals.  Deterministic in (seed, key); returns a new dict."""
    out = dict(sd)

    def pick(key, n):
        g = torch.Generator().manual_seed((seed * 1000003 + W.zlib.crc32(key.encode())) % (2 ** 31))
        idx = torch.randperm(n, generator=g)[:n_out]
        fac = lo + (hi - lo) * torch.rand(n_out, generator=g)
        return idx, fac

    for k, v in sd.items():
        if k.startswith("token_critic.net."):
            continue
        gain = k.endswith((".ln1.weight", ".ln2.weight", "ln_f.weight", ".norm.gamma", ".2.0.gamma", ".2.3.gamma")) or (".norm" in k and k.endswith(".weight") and v.dim() == 1)
        rows = k.endswith((".mlp.0.weight", ".mlp.0.bias", ".2.1.weight", ".nin_shortcut.weight", ".nin_shortcut.bias", "decoder.conv_in.weight", "decoder.conv_in.bias"))
        if k.startswith(("transformer.norm.", "transformer.self_cond_to_init_embed.")) or not (gain or rows) or not v.is_floating_point():
            continue
        key = k[:-5] + ".weight" if k.endswith(".bias") else k   # a bias shares the channels of its weight
        idx, fac = pick(key, v.shape[0])
        t = v.clone()
        t[idx] = t[idx] * fac.reshape(-1, *([1] * (t.dim() - 1))).to(t.dtype)
        out[k] = t
    for k in sd:
        if k.startswith("token_critic.net."):
            out[k] = out["transformer." + k[len("token_critic.net."):]]
    return out


def case_state_dict(case: Case, cfg):
    """Weights of a case: the deterministic reference-initialisation weights, heavy-tailed when the case says so."""
    sd = (maskgit_state_dict if case.route == "m" else gpt_state_dict)(cfg, case.weight_seed)
    return heavy_tail(sd, case.heavy) if case.heavy else sd


def inputs(case: Case, cfg):
    return synthetic.make_batch(cfg, case.batch, seed=case.input_seed)


def maskgit_noise(case: Case, cfg, seed: int = 5):
    rows = case.batch * cfg.num_cams
    T, V, ts = cfg.num_cam_tokens, cfg.vocab_size, case.timesteps
    return {"gumbel_u": synthetic.uniform_noise((ts, rows, T, V), seed, 0), "critic_u": synthetic.uniform_noise((ts, rows, T), seed, 1)}


VQ_TINY = dict(dd=presets.VQ_DDCONFIG_TINY, n_embed=64, embed_dim=64, seed=99, n_images=3)
# the released first-stage decoder at full size (f16: ch 128, ch_mult [1,1,2,2,4], 256x256, codebook 1024 x 256, configs/model/stage_2.yaml:36-55)
VQ_FULL = dict(dd=presets.VQ_DDCONFIG_F16, n_embed=1024, embed_dim=256, seed=99, n_images=1)
VQ_TINY_HEAVY = dict(dd=presets.VQ_DDCONFIG_TINY, n_embed=64, embed_dim=64, seed=99, n_images=2, heavy=41)   # vq_tiny with heavy_tail (nin_shortcut / conv_in / norm gains x 30-100):
# the un-normalised residual tensors between its two nin_shortcut convolutions pass 65504 - the f16x3 mode must REFUSE this checkpoint (BEVGEN_STATUS_F16_RANGE), fp32 runs it
VQ_TINY_HEAVY_MILD = dict(dd=presets.VQ_DDCONFIG_TINY, n_embed=64, embed_dim=64, seed=99, n_images=2, heavy=43, lo=6.0, hi=16.0)   # outliers x 6-16: inside the f16 range, pixels within 1e-3
VQ_TINY_SEG = dict(dd=dict(presets.VQ_DDCONFIG_TINY, in_channels=7, out_ch=7), n_embed=64, embed_dim=64, seed=77, n_images=2)  # BEV cond stage (7 Argoverse classes)


def golden_state_dict(case: Case, cfg, g):
    """Weights of a Route A case; when the golden carries the per-layer layouts the imported reference drew (density < 1), they replace the
    configuration's layout in the `master_layout` buffers."""
    import numpy as np

    sd = case_state_dict(case, cfg)
    if "layer_layout_bits" in g.files:
        shape = tuple(int(v) for v in g["layer_layout_shape"])
        lay = np.unpackbits(g["layer_layout_bits"])[: int(np.prod(shape))].reshape(shape).astype(np.int64)
        sd = with_layer_layouts(sd, torch.from_numpy(lay))
    return sd
