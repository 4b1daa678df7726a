"""Drop-in for multi_view_generation/modules/transformer/mingpt_sparse.py (Route A).

  GPTConfig   gpt:26-113   -> bevgen_amd.config.GPTConfig (``_target_`` at configs/model/stage_2.yaml:9)
  GPT         gpt:267-391  (``_target_`` at configs/model/stage_2.yaml:7)

``GPT`` holds the parameters under the reference's names (incl. the ``master_layout`` buffers of every block); its arithmetic runs in
libbevgen_hip as prefill + KV-cache decode (bevgen_ar_prefill / bevgen_ar_decode_step / bevgen_ar_sample).
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from ... import tables
from ... import weights as W
from ...config import Cameras, Dataset, GPTConfig  # noqa: F401  (re-exported for the YAML `_target_`)
from ...runtime import Context
from ...tables import generate_grid, get_bev_grid  # noqa: F401
from ..options import RuntimeOptionsMixin
from ..params import build_tree, module_device


class GPT(RuntimeOptionsMixin, nn.Module):
    """``GPT(cfg, **kwargs)`` (gpt:270).  Keys of ``kwargs`` this drop-in understands (next to the ``_target_`` in configs/model/stage_2.yaml:7-9, or
    ``+model.transformer.kv_cache=f16`` on the command line): ``precision``, ``weights``, ``kv_cache``, ``decode_weights``, ``decode_path`` - see
    bevgen_amd/modules/options.py; unset keys fall back to $BEVGEN_* and then to the bit-exact defaults (fp32 KV cache, fp32 decode weights)."""

    def __init__(self, cfg: GPTConfig, **kwargs):
        super().__init__()
        self._ctx: Optional[Context] = None
        self._init_runtime_options(kwargs)
        self.cfg = cfg
        if cfg.hidden_size // cfg.num_heads != 64 or cfg.hidden_size != cfg.num_embed:
            raise ValueError("libbevgen_hip needs hidden_size == num_embed and hidden_size / num_heads == 64")
        build_tree(self, W.gpt_shapes(cfg))
        # reference init (gpt:310-317): Linear/Embedding N(0, 0.02), biases 0, LayerNorm 1/0, positional / bias tables 0
        for name, p in self.named_parameters():
            if name.endswith("ln1.weight") or name.endswith("ln2.weight") or name == "ln_f.weight":
                p.data.fill_(1.0)
            elif p.dim() >= 2 and not name.endswith("pos_emb") and name not in ("camera_bias_emb", "bev_cam_pos_emb"):
                p.data.normal_(0.0, 0.02)
        if cfg.bev_embed:
            self.bev_grid.copy_(tables.get_bev_grid(cfg))
        for name, b in self.named_buffers():
            if name.endswith("master_layout"):
                b.copy_(cfg.layout)

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
                raise RuntimeError("GPT must live on a ROCm device before sampling; libbevgen_hip has no CPU path")
            ctx = Context(self.cfg, route="ar", device=dev.index if dev.index is not None else torch.cuda.current_device(), **self.runtime_options("ar"))
            # every blocks.{i}...master_layout buffer travels as int64: the reference draws one layout PER attention layer when density < 1
            # (gpt:176, maskgen:217-251) and finalize_ar reads each layer's own buffer (csrc/context.cpp); table.layout only backs absent ones
            sd = self.state_dict()
            ctx.load_state_dict({k: (v.to(torch.int64) if k.endswith("master_layout") else v) for k, v in sd.items()})
            ctx.set_tables(self.cfg)
            ctx.finalize()
            self._ctx = ctx
        return self._ctx

    @torch.no_grad()
    def forward(self, cam_indices, bev_indices, batch, sampling, **kwargs):
        """gpt:319-391: logits [B, N, V] in camera-major order for the given tokens (teacher forced).  The full L-token forward is
        evaluated as prefill + N single-row decode steps, which yields the same rows (image rows are causal in decode order)."""
        cfg = self.cfg
        if not sampling:
            cam_indices = cam_indices.clone()
            cam_indices[:, -1, -1] = cfg.vocab_size  # gpt:328-329
        ctx = self.context()
        B = cam_indices.shape[0]
        N, T = cfg.num_img_tokens, cfg.num_cam_tokens
        flat = cam_indices.reshape(B, N).to(ctx.device)
        ctx.ar_prefill(bev_indices, batch["intrinsics_inv"], batch["extrinsics_inv"])
        rows = []
        for s in range(N):
            rows.append(ctx.ar_logits())
            if s + 1 < N:
                ctx.ar_decode_step(flat[:, int(cfg.forward_shuffle_idx[s])])
        ctx.synchronize()   # gpt:383,388 (finite inputs / logits asserted by the reference): the device status word of the N step calls, read once
        logits = torch.stack(rows, dim=1)  # decode order
        return logits[:, cfg.backward_shuffle_idx.to(logits.device)]
