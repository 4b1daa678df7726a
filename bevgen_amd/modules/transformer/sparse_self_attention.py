"""Drop-in for multi_view_generation/modules/transformer/sparse_self_attention.py (the operator seam of Route A).

``SparseSelfAttention.forward(query, key, value, rpe=None, key_padding_mask=None, attn_mask=None, add_mask=None)``  (ssa:103-177)
with DeepSpeed 0.7.4's sdd -> (+add_mask) -> softmax(scale, attn_mask 'mul') -> dsd semantics, executed by the flash kernel of
libbevgen_hip (bevgen_sparse_self_attention).  Inputs may be fp16 (as DeepSpeed requires) or fp32; arithmetic is fp32.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from ...runtime import Context


class SparsityConfig:
    """Minimal stand-in for deepspeed.ops.sparse_attention.SparsityConfig (only what SparseSelfAttention reads)."""

    def __init__(self, num_heads, block=16, different_layout_per_head=False):
        self.num_heads, self.block, self.different_layout_per_head = num_heads, block, different_layout_per_head
        self.layout = None

    def make_layout(self, seq_len):
        if self.layout is not None:
            return self.layout
        n = seq_len // self.block
        return torch.ones((self.num_heads, n, n), dtype=torch.int64)


class CustomSparsityConfig(SparsityConfig):
    """gpt:143-154: hands a precomputed layout through ``make_layout``."""

    def __init__(self, num_heads, layout, block, different_layout_per_head=True):
        super().__init__(num_heads, block, different_layout_per_head)
        self.layout = layout


class SparseSelfAttention(nn.Module):
    def __init__(self, sparsity_config=None, key_padding_mask_mode="add", attn_mask_mode="mul", max_seq_length=2048):
        super().__init__()
        self.sparsity_config = sparsity_config if sparsity_config is not None else SparsityConfig(num_heads=4)
        self.register_buffer("master_layout", self.sparsity_config.make_layout(max_seq_length))
        self.key_padding_mask_mode, self.attn_mask_mode = key_padding_mask_mode, attn_mask_mode
        self._ctx: Optional[Context] = None

    def get_layout(self, L):
        if L % self.sparsity_config.block != 0:
            raise ValueError(f"Sequence Length, {L}, needs to be dividable by Block size {self.sparsity_config.block}!")
        nb = L // self.sparsity_config.block
        return self.master_layout[..., :nb, :nb]

    @torch.no_grad()
    def forward(self, query, key, value, rpe=None, key_padding_mask=None, attn_mask=None, add_mask=None):
        if rpe is not None or key_padding_mask is not None:
            raise NotImplementedError("rpe / key_padding_mask are never passed by the reference (gpt:203-207)")
        if self.attn_mask_mode != "mul":
            raise NotImplementedError("only attn_mask_mode='mul' (the mode the reference constructs, gpt:177) is implemented")
        if query.shape != key.shape or key.shape != value.shape:
            raise NotImplementedError("only self-attention is supported for now")
        B, H, L, dh = query.shape
        if self._ctx is None:
            self._ctx = Context(None, device=query.device.index if query.device.index is not None else torch.cuda.current_device())
        layout = self.get_layout(L).contiguous()
        if attn_mask is None:
            attn_mask = torch.ones((L, L), dtype=torch.float32, device=query.device)
        out = self._ctx.sparse_self_attention(query.float(), key.float(), value.float(), layout, attn_mask.float().squeeze(), add_mask, self.sparsity_config.block)
        return out.to(query.dtype)
