"""Drop-in for multi_view_generation/modules/stage2/cond_transformer_multi_view_muse.py (Route M LightningModule).

``Net2NetTransformer(maskgit, first_stage, cond_stage, cfg, ...)``  (muse_lm:29-91; ``_target_`` at
configs/experiment/muse_stage_two_multi_view.yaml:17) with ``forward(batch) -> {'gen','rec','gt'}`` (muse_lm:225-286).

What runs where:
  * sample()          -> MaskGit.generate inside libbevgen_hip (HIP kernels)
  * decode_to_img()   -> fused codebook lookup + VQGAN decoder + denormalise inside libbevgen_hip
  * 'gt'              -> util.denormalize_tensor of the input images (one fused elementwise expression; no model arithmetic)
  * encode_to_c / encode_to_z -> HIP VQGAN encoder + arg-min quantizer (bevgen_vq_encode); precomputed ``batch['cond_ids']`` /
    ``batch['z_ids']`` short-circuit them.
"""
from __future__ import annotations

import logging
import time
from typing import Optional

import torch
import torch.nn as nn

from ...modules.stage1.vqgan import VQModel
from ..options import pop_runtime_options
from .muse_maskgit_pytorch import MaskGit

log = logging.getLogger(__name__)

try:  # keep the LightningModule base when Lightning is installed so that `trainer.test(model, datamodule)` (generate.py:62) works unchanged
    import pytorch_lightning as pl

    _Base = pl.LightningModule
except Exception:  # pragma: no cover - Lightning is absent from the build image
    _Base = nn.Module

DENORM_MEAN = (0.4265, 0.4489, 0.4769)
DENORM_STD = (0.2053, 0.2206, 0.2578)


def denormalize_tensor(x: torch.Tensor) -> torch.Tensor:
    """bev_utils/util.py:97-118 (keep_tensor=True): per-channel x*std+mean, clamp [0,1]; [B,3,H,W]."""
    mean = torch.tensor(DENORM_MEAN, device=x.device, dtype=x.dtype).view(1, 3, 1, 1)
    std = torch.tensor(DENORM_STD, device=x.device, dtype=x.dtype).view(1, 3, 1, 1)
    return torch.clamp(x * std + mean, 0, 1)


class Net2NetTransformer(_Base):
    def __init__(self, maskgit: MaskGit, first_stage: VQModel, cond_stage, cfg, permuter=None, ckpt_path=None, ignore_keys=(), unfrozen_keys=(),
                 first_stage_key="image", cond_stage_key="segmentation", downsample_cond_size=-1, pkeep=1.0, sos_token=0, unconditional=False,
                 skip_sampling: bool = False, bbox_ce_weight: float = 0.0, reset_random_mask: int = 0, debug_viz: bool = False,
                 partial_decoding: Optional[int] = None, bbox_warmup_steps: int = -1, top_k: Optional[int] = None, warmup_steps: int = 500,
                 lr_decay: bool = False, sample_iterations: int = 18, **kwargs):
        super().__init__()
        runtime = pop_runtime_options(kwargs)     # precision / weights: handed down to the modules that own a Context (bevgen_amd/modules/options.py)
        for k, v in kwargs.items():
            if k != "self":
                setattr(self, k, v)
        if permuter is not None:
            raise NotImplementedError("a non-identity stage-1 permuter is not used by any shipped configuration (muse_lm:82-85)")
        if downsample_cond_size > -1:
            raise NotImplementedError("downsample_cond_size > -1 (muse_lm:149-152: F.interpolate of the condition before the cond-stage encoder) is set by no shipped "
                                      "configuration and is not implemented; refusing rather than ignoring it")
        self.first_stage_key, self.cond_stage_key = first_stage_key, cond_stage_key
        self.skip_sampling, self.partial_decoding, self.top_k = skip_sampling, partial_decoding, top_k
        self.sample_iterations = sample_iterations
        self.debug_viz = False  # wandb histogram logging of the reference is not part of the path
        self.first_stage_model = first_stage.eval()
        self.cond_stage_model = cond_stage.eval() if cond_stage is not None else None
        self.cfg = cfg
        self.maskgit = maskgit
        if runtime:
            self.set_runtime_options(_inherit=True, **runtime)
        if ckpt_path is not None:
            from ...checkpoint import init_from_ckpt

            init_from_ckpt(self, ckpt_path, ignore_keys=list(ignore_keys), unfrozen_keys=list(unfrozen_keys))

    # ------------------------------------------------------------------------------------ helpers (muse_lm:103-108)
    def expand_all_images(self, arr):
        return arr.reshape(-1, self.cfg.num_cams, *arr.shape[1:])

    def combine_all_images(self, arr):
        return arr.reshape(-1, *arr.shape[2:])

    def set_runtime_options(self, _inherit: bool = False, **opts):
        """Modes of the HIP library for the modules this one owns (bevgen_amd/modules/options.py); keys a sub-module was given itself win at construction."""
        opts = {k: v for k, v in opts.items() if k in ("precision", "weights")}
        for m in (self.maskgit, self.first_stage_model, self.cond_stage_model):
            if m is not None and opts:
                m.set_runtime_options(inherit=_inherit, **opts)

    def load_state_dict(self, *a, **k):
        out = super().load_state_dict(*a, **k)
        self.maskgit.invalidate()
        self.first_stage_model.invalidate()
        return out

    # ------------------------------------------------------------------------------------ reference API
    @torch.no_grad()
    def sample(self, cond, batch, partial_decoding_idx=None, noise=None):
        """muse_lm:120-140 -> ids [(B*C), h, w]."""
        init_ids = None
        if partial_decoding_idx is not None:
            if "z_ids" in batch:
                z = batch["z_ids"].to(cond.device)
            else:
                _, z = self.encode_to_z(self.get_input(self.first_stage_key, batch).to(cond.device), batch)
                z = z.reshape(cond.shape[0], self.cfg.num_cams, -1)
            init_ids = torch.full_like(z, self.maskgit.mask_id)
            init_ids[:, partial_decoding_idx, :] = z[:, partial_decoding_idx]
            init_ids = init_ids.reshape(-1, z.shape[-1])
        ids = self.maskgit.generate(init_ids=init_ids, cond_images=cond, fmap_size=(self.cfg.cam_latent_h, self.cfg.cam_latent_w), batch=batch,
                                    timesteps=self.sample_iterations, noise=noise)
        return ids

    @torch.no_grad()
    def decode_to_img(self, index, zshape=None, denormalize=False):
        """muse_lm:157-164 (+ denormalize_tensor when asked): ids [(B*C), T or h,w] -> [(B*C), 3, H, W]."""
        return self.first_stage_model.decode_ids(index.reshape(index.shape[0], -1), denormalize=denormalize, latent_hw=(self.cfg.cam_latent_h, self.cfg.cam_latent_w))

    @torch.no_grad()
    def encode_to_c(self, c, batch):
        """muse_lm:149-155: BEV segmentation [B, S, 256, 256] -> condition token ids [B, K] (precomputed batch['cond_ids'] short-circuits)."""
        if "cond_ids" in batch:
            return None, batch["cond_ids"]
        quant_c, _, info = self.cond_stage_model.encode(c, batch)
        return quant_c, info[2].view(c.shape[0], -1)

    @torch.no_grad()
    def encode_to_z(self, x, batch):
        """muse_lm:142-147: images [(B*C), 3, H, W] -> z ids [(B*C), T]."""
        quant_z, _, info = self.first_stage_model.encode(x, batch)
        return quant_z, info[2].view(x.shape[0], -1)

    def get_input(self, key, batch):
        x = batch[key]
        if x.dtype in (torch.double, torch.uint8):
            x = x.float()
        x = x.movedim(-1, -3)  # '... h w c -> ... c h w'
        if key == "image":
            if x.dim() == 4:
                x = x[None]
            x = self.combine_all_images(x)
        return x.contiguous()

    @torch.no_grad()
    def log_images(self, batch, generate_only=False, noise=None, **kwargs):
        """muse_lm:230-286 -> {'gen','rec','gt'} each [B, C, 3, H, W] in [0,1] ('rec'/'gt' None when their inputs are absent)."""
        start = time.time()
        dev = next(self.maskgit.parameters()).device
        _, c_indices = self.encode_to_c(None if "cond_ids" in batch else self.get_input(self.cond_stage_key, batch).to(dev), batch)
        c_indices = c_indices.to(dev)
        B = c_indices.shape[0]
        batch = dict(batch)
        batch["intrinsics_inv"] = batch["intrinsics_inv"].to(dev)
        batch["extrinsics_inv"] = batch["extrinsics_inv"].to(dev)
        partial_idx = None
        if self.partial_decoding:
            if self.partial_decoding == 2:
                partial_idx = torch.randint(self.cfg.num_cams, (torch.randint(1, self.cfg.num_cams, ()).item(),))
            elif self.partial_decoding == 3:
                partial_idx = torch.tensor([0]) if torch.rand(()).item() > 0.5 else torch.tensor([0, 2])
            else:
                partial_idx = torch.randint(self.cfg.num_cams, (1,))
        index_sample = self.sample(c_indices, batch, partial_decoding_idx=partial_idx, noise=noise)
        assert index_sample.max() < self.cfg.vocab_size
        gen = self.expand_all_images(self.decode_to_img(index_sample, denormalize=True))
        rec, gt = None, None
        x_img = self.get_input(self.first_stage_key, batch).to(dev) if self.first_stage_key in batch else None
        if "z_ids" in batch:
            z = batch["z_ids"].to(dev).reshape(B * self.cfg.num_cams, -1)
        elif x_img is not None:
            _, z = self.encode_to_z(x_img, batch)           # reconstruction path of the reference (muse_lm:240-250)
        else:
            z = None
        if z is not None:
            rec = self.expand_all_images(self.decode_to_img(z, denormalize=True))
        if x_img is not None:
            gt = self.expand_all_images(denormalize_tensor(x_img))
        log.info("Generating images took %.3f s", time.time() - start)
        return {"gen": gen, "rec": rec, "gt": gt}

    def test_step(self, batch, batch_idx):
        return self.log_images(batch, generate_only=True)

    def forward(self, batch):
        return self.log_images(batch, generate_only=True)
