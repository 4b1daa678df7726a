"""Drop-in for multi_view_generation/modules/stage1/vqgan.py (+ the parts of quantize.py / model.py it instantiates).

  VQModel                 stage1/vqgan.py:31-213     (``_target_`` at configs/model/stage_2.yaml:36)
  VQSegmentationModel     stage1/vqgan.py:216-261    (``_target_`` at configs/model/stage_2.yaml:59)
  VectorQuantizer2        stage1/quantize.py:213-329 (``quantize`` attribute; ``get_codebook_entry`` is the decode-side entry)

Parameters carry the reference's ``state_dict`` names (encoder.*, decoder.*, quantize.embedding.weight, quant_conv.*, post_quant_conv.*).
``decode`` runs the hand-written HIP decoder (bevgen_vq_decode / bevgen_vq_decode_latents); ``encode`` (the step BEFORE the path)
runs the HIP encoder + arg-min quantizer (bevgen_vq_encode).
"""
from __future__ import annotations

from typing import Mapping, Optional

import torch
import torch.nn as nn

from ... import weights as W
from ...runtime import Context
from ..options import RuntimeOptionsMixin
from ..params import ParamNode, build_tree, module_device


class VectorQuantizer2(ParamNode):
    def __init__(self, n_e, e_dim, beta=0.25, remap=None, unknown_index="random", sane_index_shape=False, legacy=True):
        super().__init__()
        if remap is not None:
            raise NotImplementedError("index remapping is not used by any shipped configuration")
        self.n_e, self.e_dim, self.beta, self.legacy = n_e, e_dim, beta, legacy
        self.re_embed = n_e
        self.sane_index_shape = sane_index_shape

    def get_codebook_entry(self, indices, shape):
        """quant:314-329 - z_q = embedding(indices).view(b,h,w,c).permute(0,3,1,2) (API-compatible helper; the fused path is VQModel.decode_ids)."""
        z_q = self.embedding.weight[indices]
        if shape is not None:
            z_q = z_q.view(shape).permute(0, 3, 1, 2).contiguous()
        return z_q


class VQModel(RuntimeOptionsMixin, nn.Module):
    def __init__(self, ddconfig, lossconfig=None, n_embed=None, embed_dim=None, cam_res=None, cam_latent_res=None, cam_emd_dim=None,
                 geometric_embedding=False, ckpt_path=None, ignore_keys=(), image_key="image", colorize_nlabels=None, monitor=None, remap=None,
                 sane_index_shape=False, denormalize=True, legacy=True, **kwargs):
        super().__init__()
        self._ctx: Optional[Context] = None
        self._init_runtime_options(kwargs)    # precision / weights (bevgen_amd/modules/options.py)
        if geometric_embedding:
            raise NotImplementedError("geometric_embedding=True is not used by the released checkpoints (configs/model/stage_2_argoverse.yaml:8,11)")
        self.ddconfig = dict(ddconfig)
        self.ddconfig["ch_mult"] = list(self.ddconfig["ch_mult"])
        self.ddconfig["attn_resolutions"] = list(self.ddconfig["attn_resolutions"])
        self.n_embed, self.embed_dim = n_embed, embed_dim
        # the networks are fully convolutional: cam_res / cam_latent_res only fix the default latent grid of decode_ids (nuScenes: 224 x 400 -> 14 x 25)
        self.cam_res = tuple(cam_res) if cam_res is not None else None
        self.cam_latent_res = tuple(cam_latent_res) if cam_latent_res is not None else None
        self.image_key = image_key
        self.denormalize = denormalize
        self.quantize = VectorQuantizer2(n_embed, embed_dim, beta=0.25, remap=remap, sane_index_shape=sane_index_shape, legacy=legacy)
        shapes = W.vqmodel_shapes(self.ddconfig, n_embed, embed_dim, with_encoder=True)
        build_tree(self, shapes)
        for name, p in self.named_parameters():
            if p.dim() >= 2:
                nn.init.kaiming_uniform_(p, a=5 ** 0.5)
            elif name.endswith("weight"):
                p.data.fill_(1.0)
        if ckpt_path is not None:
            from ...checkpoint import init_from_ckpt

            init_from_ckpt(self, ckpt_path, ignore_keys=list(ignore_keys), strict=False)

    # ---------------------------------------------------------------------------------- device context
    def load_state_dict(self, *a, **k):
        out = super().load_state_dict(*a, **k)
        self.invalidate()
        return out

    def invalidate(self):
        if self._ctx is not None:
            self._ctx.close()
            self._ctx = None

    def context(self) -> Context:
        if self._ctx is None:
            dev = module_device(self)
            if dev.type != "cuda":
                raise RuntimeError("VQModel must live on a ROCm device before decode(); libbevgen_hip has no CPU path")
            ctx = Context(None, vq_ddconfig=self.ddconfig, vq_n_embed=self.n_embed, vq_embed_dim=self.embed_dim,
                          device=dev.index if dev.index is not None else torch.cuda.current_device(), **self.runtime_options("vq"))
            sd = {k: v for k, v in self.state_dict().items() if k != "colorize"}
            ctx.load_state_dict(sd, prefix="first_stage_model.")
            ctx.finalize()
            self._ctx = ctx
        return self._ctx

    # ---------------------------------------------------------------------------------- reference API
    @torch.no_grad()
    def decode(self, quant):
        """vqgan:118-121 - post_quant_conv + Decoder on looked-up latents [n, embed_dim, h, w] -> [n, out_ch, H, W]."""
        return self.context().vq_decode_latents(quant, denormalize=False)

    @torch.no_grad()
    def decode_ids(self, ids, denormalize=False, latent_hw=None):
        """Fused get_codebook_entry + decode (+ util.denormalize_tensor): ids [n, h*w] -> pixels; latent grid = latent_hw, else cam_latent_res, else square."""
        return self.context().vq_decode(ids, denormalize=denormalize, latent_hw=latent_hw or self.cam_latent_res)

    def decode_code(self, code_b):
        return self.decode_ids(code_b.reshape(code_b.shape[0], -1))

    @torch.no_grad()
    def encode_ids(self, x):
        """Encoder -> quant_conv -> arg-min over the codebook: x [n, in_channels, H, W] -> ids [n, h*w]."""
        return self.context().vq_encode(x)

    @torch.no_grad()
    def encode(self, x, batch=None):
        """vqgan:84-116 (geometric_embedding=False) -> (quant [n, e, h, w], emb_loss=None, (None, None, indices [n*h*w]))."""
        ids = self.encode_ids(x)
        f = 2 ** (len(self.ddconfig["ch_mult"]) - 1)
        quant = self.quantize.get_codebook_entry(ids.reshape(-1), (x.shape[0], x.shape[-2] // f, x.shape[-1] // f, self.embed_dim))
        return quant, None, (None, None, ids.reshape(-1))

    def forward(self, input, batch=None):
        raise NotImplementedError("stage-1 training/reconstruction forward is outside the stage-2 sampling path")


class VQSegmentationModel(VQModel):
    def __init__(self, n_labels, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.n_labels = n_labels
        self.register_buffer("colorize", torch.randn(3, n_labels, 1, 1))  # checkpoint key (vqgan:219); only used by visualisation
