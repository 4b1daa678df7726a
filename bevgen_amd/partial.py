"""Partial decoding helpers: some cameras keep the tokens of their encoded ground-truth image, the sampler fills in the rest.

Route M passes them as ``init_ids`` (muse_lm:124-132).  Route A (ar_lm:161-165, 181-182) initialises the fixed cameras' positions and skips them in
the decode loop; with the KV cache every position still has to be pushed through the stack in decode order, so the fixed positions become FORCED
tokens of ``bevgen_ar_sample_forced``: they are emitted instead of a drawn token and fed back like any other.
"""
from __future__ import annotations

from typing import Iterable, Optional

import torch


def partial_forced_ids(cfg, partial_decoding_idx: Iterable[int], z_indices: torch.Tensor, steps: Optional[int] = None) -> torch.Tensor:
    """-> [steps, B] int64 in decode order: z_indices[b, cam, pos] where the step's camera is fixed, -1 elsewhere.
    Step s visits flat token j = cfg.forward_shuffle_idx[s] = camera j // T, position j % T (perm:33-88)."""
    T = cfg.num_cam_tokens
    n = cfg.num_img_tokens if steps is None else int(steps)
    z = z_indices.reshape(z_indices.shape[0], cfg.num_cams * T)
    order = torch.as_tensor(cfg.forward_shuffle_idx[:n], dtype=torch.long, device=z.device)
    fixed = torch.zeros(cfg.num_cams, dtype=torch.bool, device=z.device)
    fixed[torch.as_tensor(list(partial_decoding_idx), dtype=torch.long, device=z.device)] = True
    out = z[:, order].t().contiguous()                      # [n, B]
    out[~fixed[order // T]] = -1
    return out
