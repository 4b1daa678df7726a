"""Build the *imported reference* modules with this repo's deterministic weights (build container only).

TEST INFRASTRUCTURE.  Constructors are called directly with kwargs (no Hydra), as SURVEY.md section 8c describes;
the only behavioural patch is Route A's ``SparseSelfAttention.forward`` whose DeepSpeed/Triton ops are not
installable here: it is replaced by ``oracle.restate.sparse_self_attention_dense`` (the dense restatement of the same
formula) so the rest of the reference ``GPT`` stack runs unmodified.
"""
from __future__ import annotations

import dataclasses
import os
import tempfile
from contextlib import contextmanager
from typing import Dict, Mapping

import torch

from bevgen_amd import weights as W
from oracle import restate
from . import stubs

_GPTCFG_FIELDS = ("embd_pdrop resid_pdrop attn_pdrop num_layers num_heads num_embed hidden_size vocab_size cond_vocab_size num_cams "
                  "window_len density sparse_block_size n_unmasked plot cam_res cam_latent_res bev_latent_res camera_bias bev_embed "
                  "image_embed causal_order legacy_prob_matrix").split()


@contextmanager
def _cwd_with_cam_data(cfg):
    """The reference reads pretrained/cam_data_<dataset>.pt relative to cwd (maskgen:90)."""
    old = os.getcwd()
    with tempfile.TemporaryDirectory() as d:
        os.makedirs(os.path.join(d, "pretrained"))
        if cfg.cam_intrinsics is not None:
            torch.save({"intrinsics": torch.as_tensor(cfg.cam_intrinsics)[None], "extrinsics": torch.as_tensor(cfg.cam_extrinsics)[None]},
                       os.path.join(d, "pretrained", f"cam_data_{cfg.dataset_name}.pt"))
        os.chdir(d)
        try:
            yield
        finally:
            os.chdir(old)


def ref_gpt_config(cfg):
    """Reference ``GPTConfig`` equivalent to a ``bevgen_amd.config.GPTConfig``."""
    ns = stubs.import_reference()
    kw = {k: getattr(cfg, k) for k in _GPTCFG_FIELDS}
    kw.update(cam_names=cfg.cam_names.name, dataset=cfg.dataset.name, backend="deepspeed")
    with _cwd_with_cam_data(cfg):
        return ns.gpt.GPTConfig(**kw)


from oracle.cases import muse_kwargs, maskgit_state_dict, gpt_state_dict  # noqa: E402,F401


def vq_state_dict(dd: Mapping, n_embed: int, embed_dim: int, seed: int, with_encoder: bool = True):
    from oracle.cases import vq_state_dict as _v

    return _v(dd, n_embed, embed_dim, seed, with_encoder=with_encoder)


def build_ref_maskgit(cfg, sd):
    ns = stubs.import_reference()
    rcfg = ref_gpt_config(cfg)
    tr = ns.muse_net.MaskGitTransformerMultiView(num_tokens=cfg.vocab_size, dim=cfg.num_embed, seq_len=cfg.cam_latent_res,
                                                 depth=cfg.num_layers, dim_head=64, heads=cfg.num_heads, ff_mult=4, cfg=rcfg)
    mg = ns.muse_net.MaskGit(image_size=cfg.cam_latent_res, transformer=tr, self_token_critic=True, cond_drop_prob=0.1)
    missing, unexpected = mg.load_state_dict(sd, strict=True), None
    return mg.eval(), rcfg


def build_ref_gpt(cfg, sd, layout_seed: int = 0):
    """layout_seed != 0 (density < 1): the reference GPT is constructed under that torch seed and the per-layer layouts it draws
    (one multi_outward_pattern call per attention module, gpt:176) are written into ``sd`` (in place) before loading, i.e. they survive."""
    ns = stubs.import_reference()
    rcfg = ref_gpt_config(cfg)
    if layout_seed:
        torch.manual_seed(layout_seed)
    gpt = ns.gpt.GPT(rcfg)
    if layout_seed:
        for i, blk in enumerate(gpt.blocks):
            sd[f"blocks.{i}.attention.sparse_self_attention.master_layout"] = blk.attention.sparse_self_attention.master_layout.clone().to(torch.int64)
    gpt.load_state_dict(sd, strict=True)

    def dense_forward(self, query, key, value, rpe=None, key_padding_mask=None, attn_mask=None, add_mask=None):
        return restate.sparse_self_attention_dense(query, key, value, self.master_layout, self.sparsity_config.block, attn_mask, add_mask)

    ns.ssa.SparseSelfAttention.forward = dense_forward
    return gpt.eval(), rcfg


def build_ref_vqmodel(dd: Mapping, n_embed: int, embed_dim: int, sd, cam_res, cam_latent_res):
    ns = stubs.import_reference()
    from multi_view_generation.modules.losses.vqperceptual import DummyLoss

    vq = ns.vqgan.VQModel(ddconfig=dict(dd), lossconfig=DummyLoss(), n_embed=n_embed, embed_dim=embed_dim, cam_res=cam_res,
                          cam_latent_res=cam_latent_res, cam_emd_dim=embed_dim)
    vq.load_state_dict(sd, strict=True)
    return vq.eval()


@contextmanager
def deterministic_maskgit_noise(noise=None):
    """Patch the reference's two noise sources (muse_net:430-431, 446-448).  ``noise=None``: gumbel 0, uniform 0.5 (greedy goldens);
    else feed the explicit uniforms step by step."""
    ns = stubs.import_reference()
    mn = ns.muse_net
    old_g, old_u = mn.gumbel_noise, mn.uniform
    state = {"g": 0, "u": 0}

    def gumbel_noise(t):
        if noise is None:
            return torch.zeros_like(t)
        u = noise["gumbel_u"][state["g"]]
        state["g"] += 1
        return -mn.log(-mn.log(u))

    def uniform(shape, min=0, max=1, device=None):
        if noise is None:
            return torch.full(tuple(shape), 0.5)
        u = noise["critic_u"][state["u"]]
        state["u"] += 1
        return u

    mn.gumbel_noise, mn.uniform = gumbel_noise, uniform
    try:
        yield
    finally:
        mn.gumbel_noise, mn.uniform = old_g, old_u
