"""Drop-in for multi_view_generation/modules/stage2/cond_transformer_multi_view.py (Route A LightningModule).

``Net2NetTransformer(transformer, first_stage, cond_stage, ...)`` (ar_lm:30-100; ``_target_`` at configs/model/stage_2.yaml:1) with
``sample(x, c, batch, temperature, sample, top_k, callback, partial_decoding_idx)`` (ar_lm:154-227) executed as prefill + KV-cache decode
inside libbevgen_hip, and ``top_k_logits`` (ar_lm:138-142).
"""
from __future__ import annotations

import logging
from typing import Optional

import torch
import torch.nn as nn

from .cond_transformer_multi_view_muse import _Base, denormalize_tensor

log = logging.getLogger(__name__)


class Net2NetTransformer(_Base):
    def __init__(self, transformer, first_stage, cond_stage, permuter=None, ckpt_path=None, ignore_keys=(), unfrozen_keys=(), first_stage_key="image",
                 cond_stage_key="segmentation", downsample_cond_size=-1, pkeep=1.0, sos_token=0, unconditional=False, skip_sampling: bool = False,
                 bbox_ce_weight: float = 0.0, reset_random_mask: int = 0, debug_viz: bool = False, partial_decoding: Optional[int] = None,
                 bbox_weight_epoch: int = -1, top_k: Optional[int] = None, warmup_steps: int = 500, lr_decay: bool = False, **kwargs):
        super().__init__()
        for k, v in kwargs.items():
            if k != "self":
                setattr(self, k, v)
        if permuter is not None:
            raise NotImplementedError("a non-identity stage-1 permuter is not used by any shipped configuration")
        self.first_stage_key, self.cond_stage_key = first_stage_key, cond_stage_key
        self.skip_sampling, self.partial_decoding, self.top_k = skip_sampling, partial_decoding, top_k
        self.first_stage_model = first_stage.eval() if first_stage is not None else None
        self.cond_stage_model = cond_stage.eval() if cond_stage is not None else None
        self.transformer = transformer
        self.cfg = transformer.cfg
        if ckpt_path is not None:
            from ...checkpoint import init_from_ckpt

            init_from_ckpt(self, ckpt_path, ignore_keys=list(ignore_keys), unfrozen_keys=list(unfrozen_keys))

    def load_state_dict(self, *a, **k):
        out = super().load_state_dict(*a, **k)
        self.transformer.invalidate()
        if self.first_stage_model is not None:
            self.first_stage_model.invalidate()
        return out

    def top_k_logits(self, logits, k):
        """ar_lm:138-142: values below the k-th largest -> -inf (ties kept)."""
        v, _ = torch.topk(logits, k)
        out = logits.clone()
        out[out < v[..., [-1]]] = -float("Inf")
        return out

    @torch.no_grad()
    def sample(self, x, c, batch, temperature=1.0, sample=False, top_k=None, callback=lambda k: None, partial_decoding_idx=None, noise_u=None):
        """ar_lm:154-227 -> x [B, C, T].  ``sample=True`` draws with explicit uniforms ``noise_u`` [N, B] (default: torch's device RNG)."""
        cfg = self.cfg
        B = c.shape[0]
        if self.skip_sampling:
            return torch.zeros((B, cfg.num_cams, cfg.num_cam_tokens), dtype=torch.int64, device=c.device)
        ctx = self.transformer.context()
        forced = None
        if partial_decoding_idx is not None:
            # ar_lm:161-165: the fixed cameras take the tokens of their encoded ground-truth images (precomputed batch['z_ids'] short-circuits)
            from ...partial import partial_forced_ids
            if "z_ids" in batch:
                z = batch["z_ids"].to(ctx.device)
            else:
                img = batch[self.first_stage_key].to(ctx.device).float().movedim(-1, -3)
                z = self.first_stage_model.encode_ids(img.reshape(-1, *img.shape[-3:]))
            idx = partial_decoding_idx.tolist() if isinstance(partial_decoding_idx, torch.Tensor) else list(partial_decoding_idx)
            forced = partial_forced_ids(cfg, idx, z.reshape(B, cfg.num_cams, cfg.num_cam_tokens))
        if sample and noise_u is None:
            noise_u = torch.rand((cfg.num_img_tokens, B), device=ctx.device)
        out = ctx.ar_sample(c, batch["intrinsics_inv"], batch["extrinsics_inv"], top_k=top_k, temperature=temperature, greedy=not sample, noise_u=noise_u,
                            forced_ids=forced)
        assert out.max() < cfg.vocab_size
        return out

    @torch.no_grad()
    def decode_to_img(self, index, zshape=None, denormalize=False):
        return self.first_stage_model.decode_ids(index.reshape(index.shape[0], -1), denormalize=denormalize)

    @torch.no_grad()
    def log_images(self, batch, temperature=None, top_k=None, callback=None, lr_interface=False, generate_only=False, **kwargs):
        """ar_lm:479-561 reduced to the generate path: {'gen': [B,C,3,H,W], 'rec': None, 'gt': ...}."""
        dev = next(self.transformer.parameters()).device
        c = batch["cond_ids"].to(dev) if "cond_ids" in batch else self.cond_stage_model.encode(batch[self.cond_stage_key], batch)
        b2 = dict(batch, intrinsics_inv=batch["intrinsics_inv"].to(dev), extrinsics_inv=batch["extrinsics_inv"].to(dev))
        x = self.sample(None, c, b2, temperature=temperature if temperature is not None else 1.0, sample=True, top_k=top_k if top_k is not None else 100)
        gen = self.decode_to_img(x.reshape(-1, cfg_T(self.cfg)), denormalize=True)
        gen = gen.reshape(c.shape[0], self.cfg.num_cams, *gen.shape[1:])
        gt = None
        if self.first_stage_key in batch:
            img = batch[self.first_stage_key].to(dev).float().movedim(-1, -3)
            gt = denormalize_tensor(img.reshape(-1, *img.shape[-3:])).reshape(img.shape)
        return {"gen": gen, "rec": None, "gt": gt}

    def test_step(self, batch, batch_idx):
        return self.log_images(batch, generate_only=True, top_k=self.top_k)

    def forward(self, batch):
        return self.log_images(batch, generate_only=True, top_k=self.top_k)


def cfg_T(cfg):
    return cfg.num_cam_tokens
