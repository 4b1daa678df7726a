"""Scene-parallel execution over the GPUs of one node (one process per GPU, torch.distributed backend "nccl" = RCCL over xGMI).

The reference shards generation with Lightning DDP + DistributedSampler and never exchanges results: every rank writes its own
files (configs/trainer/default.yaml:25, configs/modes/generate.yaml:17-18, utils/callback.py:46,64).  Scenes are independent, so
there is no data-path collective here either; the only exchange is one gather of the finished uint8 pixels
(1.18 MB per six-view scene) to rank 0.
"""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

import torch


def _passthrough(dist) -> bool:
    """No collective needed: no process group, or a group of one - unless $BEVGEN_FORCE_COLLECTIVE=1 asks for the collective anyway (the GPU suite's single-GPU RCCL
    test: one rank is all a 1-GPU box can host, and the call path through RCCL should still have run on hardware once)."""
    if dist is None or not dist.is_initialized():
        return True
    return dist.get_world_size() == 1 and os.environ.get("BEVGEN_FORCE_COLLECTIVE") != "1"


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [begin, end) shard of `total` scenes for `rank` (sizes differ by at most one; earlier ranks get the extras)."""
    base, extra = divmod(total, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def to_uint8(px: torch.Tensor) -> torch.Tensor:
    """[0,1] float pixels -> uint8 (round(x*255)), the wire/storage format of the generated images."""
    return (px * 255.0).round().clamp_(0, 255).to(torch.uint8)


def gather_scenes(px: torch.Tensor, dist=None, dst: int = 0) -> Optional[torch.Tensor]:
    """Gather per-rank pixel blocks [b, C, 3, H, W] (uint8 as bevgen_vq_decode's BEVGEN_VQ_OUT_U8 writes them, or float in [0,1]) as uint8 on rank
    `dst`; returns the concatenation there, None elsewhere.  All ranks must pass blocks of the same shape (pad the last shard if the scene count
    does not divide)."""
    u8 = px if px.dtype == torch.uint8 else to_uint8(px)
    if _passthrough(dist):
        return u8
    world, rank = dist.get_world_size(), dist.get_rank()
    bufs: Optional[List[torch.Tensor]] = [torch.empty_like(u8) for _ in range(world)] if rank == dst else None
    dist.gather(u8, bufs, dst=dst)
    return torch.cat(bufs, dim=0) if rank == dst else None


def gather_token_ids(ids: torch.Tensor, dist=None, dst: int = 0) -> Optional[torch.Tensor]:
    """Same for token ids (int32 on the wire, 6 KB per six-view scene: int16 is not a collective dtype on every backend)."""
    small = ids.to(torch.int32)
    if _passthrough(dist):
        return small.to(torch.int64)
    world, rank = dist.get_world_size(), dist.get_rank()
    bufs = [torch.empty_like(small) for _ in range(world)] if rank == dst else None
    dist.gather(small, bufs, dst=dst)
    return torch.cat(bufs, dim=0).to(torch.int64) if rank == dst else None
