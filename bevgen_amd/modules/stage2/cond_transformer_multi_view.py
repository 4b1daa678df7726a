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

from ..options import pop_runtime_options
from .cond_transformer_multi_view_muse import _Base, denormalize_tensor

log = logging.getLogger(__name__)


class Net2NetTransformer(_Base):
    def __init__(self, transformer, first_stage, cond_stage, permuter=None, ckpt_path=None, ignore_keys=(), unfrozen_keys=(), first_stage_key="image",
                 cond_stage_key="segmentation", downsample_cond_size=-1, pkeep=1.0, sos_token=0, unconditional=False, skip_sampling: bool = False,
                 bbox_ce_weight: float = 0.0, reset_random_mask: int = 0, debug_viz: bool = False, partial_decoding: Optional[int] = None,
                 bbox_weight_epoch: int = -1, top_k: Optional[int] = None, warmup_steps: int = 500, lr_decay: bool = False, **kwargs):
        super().__init__()
        runtime = pop_runtime_options(kwargs)     # precision / weights / kv_cache / decode_weights / decode_path: handed down to the modules that own a Context
        for k, v in kwargs.items():
            if k != "self":
                setattr(self, k, v)
        if permuter is not None:
            raise NotImplementedError("a non-identity stage-1 permuter is not used by any shipped configuration")
        if downsample_cond_size > -1:
            raise NotImplementedError("downsample_cond_size > -1 (ar_lm:253-257: F.interpolate of the condition before the cond-stage encoder) is set by no shipped "
                                      "configuration and is not implemented; refusing rather than ignoring it")
        self.first_stage_key, self.cond_stage_key = first_stage_key, cond_stage_key
        self.skip_sampling, self.partial_decoding, self.top_k = skip_sampling, partial_decoding, top_k
        self.first_stage_model = first_stage.eval() if first_stage is not None else None
        self.cond_stage_model = cond_stage.eval() if cond_stage is not None else None
        self.transformer = transformer
        self.cfg = transformer.cfg
        if runtime:
            self.set_runtime_options(_inherit=True, **runtime)
        if ckpt_path is not None:
            from ...checkpoint import init_from_ckpt

            init_from_ckpt(self, ckpt_path, ignore_keys=list(ignore_keys), unfrozen_keys=list(unfrozen_keys))

    def set_runtime_options(self, _inherit: bool = False, **opts):
        """Modes of the HIP library for the modules this one owns (bevgen_amd/modules/options.py); keys a sub-module was given itself win at construction."""
        self.transformer.set_runtime_options(inherit=_inherit, **opts)
        stage1 = {k: v for k, v in opts.items() if k in ("precision", "weights")}
        for m in (self.first_stage_model, self.cond_stage_model):
            if m is not None and stage1:
                m.set_runtime_options(inherit=_inherit, **stage1)

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

    # ---- helpers shared with the reference's get_input / encode_to_* (ar_lm:229-292)
    def expand_all_images(self, arr):
        return arr.reshape(-1, self.cfg.num_cams, *arr.shape[1:])

    def combine_all_images(self, arr):
        return arr.reshape(-1, *arr.shape[2:])

    def get_input(self, key, batch):
        """ar_lm:276-287: channel-last batch tensors -> channel-first; images [B, C, H, W, 3] -> [(B C), 3, H, W]."""
        x = batch[key]
        if x.dtype in (torch.double, torch.uint8):
            x = x.float()
        x = x.movedim(-1, -3)
        if key == self.first_stage_key:
            if x.dim() == 4:
                x = x[None]
            x = self.combine_all_images(x)
        return x.contiguous()

    @torch.no_grad()
    def encode_to_c(self, c, batch):
        """ar_lm:253-264: BEV segmentation -> condition token ids [B, K] (precomputed batch['cond_ids'] short-circuits)."""
        if "cond_ids" in batch:
            return None, batch["cond_ids"]
        quant_c, _, info = self.cond_stage_model.encode(c, batch)
        return quant_c, info[2].view(c.shape[0], -1)

    @torch.no_grad()
    def encode_to_z(self, x, batch):
        """ar_lm:229-242: images [(B C), 3, H, W] -> z ids [(B C), T]."""
        quant_z, _, info = self.first_stage_model.encode(x, batch)
        return quant_z, info[2].view(x.shape[0], -1)

    @torch.no_grad()
    def decode_to_img(self, index, zshape=None, denormalize=False):
        """ar_lm:244-251 with zshape = cam_latent_res (the decoder is fully convolutional: 14 x 25 latents for nuScenes)."""
        return self.first_stage_model.decode_ids(index.reshape(index.shape[0], -1), denormalize=denormalize, latent_hw=(self.cfg.cam_latent_h, self.cfg.cam_latent_w))

    def _partial_decoding_idx(self):
        """ar_lm:501-514."""
        if not self.partial_decoding:
            return None
        C = self.cfg.num_cams
        if self.partial_decoding == 2:
            return torch.randint(C, (torch.randint(1, 3, ()).item(),))
        if self.partial_decoding == 3:
            return torch.tensor([0]) if torch.rand(()).item() > 0.5 else torch.tensor([0, 2])
        if self.partial_decoding == 4:
            return torch.tensor([3, 0, 2])
        return torch.randint(C, (1,))

    @torch.no_grad()
    def log_images(self, batch, temperature=None, top_k=None, callback=None, lr_interface=False, generate_only=False, noise_u=None, sample=True, **kwargs):
        """ar_lm:479-561 reduced to what the generate path returns: {'gen', 'rec', 'gt'} each [B, C, 3, H, W] in [0,1] ('rec' / 'gt' None when the
        ground-truth images are absent).  Draws with top-k 100 like the reference unless told otherwise; `partial_decoding` picks the fixed cameras
        exactly as ar_lm:501-514 does and feeds their encoded ground-truth tokens to `sample`."""
        dev = next(self.transformer.parameters()).device
        batch = dict(batch)
        _, c = self.encode_to_c(None if "cond_ids" in batch else self.get_input(self.cond_stage_key, batch).to(dev), batch)
        c = c.to(dev)
        B = c.shape[0]
        batch["intrinsics_inv"] = batch["intrinsics_inv"].to(dev)
        batch["extrinsics_inv"] = batch["extrinsics_inv"].to(dev)
        x_img = self.get_input(self.first_stage_key, batch).to(dev) if self.first_stage_key in batch else None
        z = None
        if "z_ids" in batch:
            z = batch["z_ids"].to(dev).reshape(B * self.cfg.num_cams, -1)
        elif x_img is not None:
            _, z = self.encode_to_z(x_img, batch)
        pidx = self._partial_decoding_idx()
        if pidx is not None:
            if z is None:
                raise ValueError("partial_decoding needs the ground-truth images (batch['image']) or their token ids (batch['z_ids'])")
            batch["z_ids"] = z.reshape(B, self.cfg.num_cams, -1)
        x = self.sample(None, c, batch, temperature=temperature if temperature is not None else 1.0, sample=sample, top_k=top_k if top_k is not None else 100,
                        partial_decoding_idx=pidx, noise_u=noise_u)
        gen = self.expand_all_images(self.decode_to_img(self.combine_all_images(x), denormalize=True))
        rec = self.expand_all_images(self.decode_to_img(z, denormalize=True)) if z is not None else None
        gt = self.expand_all_images(denormalize_tensor(x_img)) if x_img is not None else None
        return {"gen": gen, "rec": rec, "gt": gt}

    def test_step(self, batch, batch_idx):
        return self.log_images(batch, generate_only=True, top_k=self.top_k)

    def forward(self, batch):
        return self.log_images(batch, generate_only=True, top_k=self.top_k)

