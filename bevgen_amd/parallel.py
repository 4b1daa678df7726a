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


class GatherMismatch(RuntimeError):
    """A gathered block does not carry the checksum its sender computed."""


def payload_checksum(t: torch.Tensor) -> torch.Tensor:
    """Two int64 words of a payload block, computed where the block lives: the plain sum and a position-weighted sum (weights 1 .. 8191 cyclic) - a transposed, shifted or
    zero-filled block changes at least one of them.  Sent beside the payload; the receiver recomputes them on what arrived."""
    flat = t.contiguous().reshape(-1)
    if flat.dtype == torch.uint8 and flat.numel() % 4 == 0:
        flat = flat.view(torch.int32)          # four pixels per word: a quarter of the elements to widen (18.9 MB of pixels per sixteen-scene block)
    v = flat.to(torch.int64)
    w = (torch.arange(v.numel(), device=v.device, dtype=torch.int64) % 8191) + 1
    return torch.stack([v.sum(), (v * w).sum()])


def _gather_verified(block: torch.Tensor, dist, dst: int, verify: bool) -> Optional[List[torch.Tensor]]:
    world, rank = dist.get_world_size(), dist.get_rank()
    bufs: Optional[List[torch.Tensor]] = [torch.empty_like(block) for _ in range(world)] if rank == dst else None
    dist.gather(block, bufs, dst=dst)
    if verify:
        # the sender's checksum travels as a second, 16-byte gather; rank `dst` recomputes it on the received blocks (one pass over what it is about to hand on anyway)
        mine = payload_checksum(block)
        sums = [torch.empty_like(mine) for _ in range(world)] if rank == dst else None
        dist.gather(mine, sums, dst=dst)
        if rank == dst:
            got = torch.stack([payload_checksum(b) for b in bufs])
            want = torch.stack(sums)
            if not torch.equal(got, want):
                bad = [r for r in range(world) if not torch.equal(got[r], want[r])]
                raise GatherMismatch(f"gather to rank {dst}: the blocks of ranks {bad} do not carry the checksums their senders computed")
    return bufs


def gather_scenes(px: torch.Tensor, dist=None, dst: int = 0, verify: bool = True) -> Optional[torch.Tensor]:
    """Gather per-rank pixel blocks [b, C, 3, H, W] (uint8 as bevgen_vq_decode's BEVGEN_VQ_OUT_U8 writes them, or float in [0,1]) as uint8 on rank
    `dst`; returns the concatenation there, None elsewhere.  All ranks must pass blocks of the same shape (pad the last shard if the scene count
    does not divide).  verify: every rank sends a checksum of its block beside it and rank `dst` checks what arrived (GatherMismatch)."""
    u8 = px if px.dtype == torch.uint8 else to_uint8(px)
    if _passthrough(dist):
        return u8
    bufs = _gather_verified(u8.contiguous(), dist, dst, verify)
    return torch.cat(bufs, dim=0) if dist.get_rank() == dst else None


def gather_token_ids(ids: torch.Tensor, dist=None, dst: int = 0, verify: bool = True) -> Optional[torch.Tensor]:
    """Same for token ids (int32 on the wire, 6 KB per six-view scene: int16 is not a collective dtype on every backend)."""
    small = ids.to(torch.int32)
    if _passthrough(dist):
        return small.to(torch.int64)
    bufs = _gather_verified(small.contiguous(), dist, dst, verify)
    return torch.cat(bufs, dim=0).to(torch.int64) if dist.get_rank() == dst else None


def bind_to_gpu_numa_node(device_index: int) -> Optional[dict]:
    """Pin this process (one rank = one GPU) to the CPUs of the NUMA node its GPU hangs off: at 0.8 ms per replayed decode step, a host thread that wanders to the other
    socket shows up as launch jitter.  PCI address from torch's device properties -> /sys/bus/pci/devices/<addr>/numa_node -> /sys/devices/system/node/node<N>/cpulist ->
    os.sched_setaffinity (intersected with the CPUs this process may already use).  Returns what it did, or None when the topology is not exposed (containers, one node)."""
    try:
        p = torch.cuda.get_device_properties(device_index)
        addr = f"{getattr(p, 'pci_domain_id', 0):04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
        with open(f"/sys/bus/pci/devices/{addr}/numa_node") as f:
            node = int(f.read().strip())
        if node < 0:
            return None
        with open(f"/sys/devices/system/node/node{node}/cpulist") as f:
            cpus = parse_cpulist(f.read())
        allowed = os.sched_getaffinity(0)
        use = sorted(set(cpus) & set(allowed))
        if not use or len(use) == len(allowed):
            return {"pci": addr, "numa_node": node, "cpus": len(use), "bound": False}
        os.sched_setaffinity(0, use)
        return {"pci": addr, "numa_node": node, "cpus": len(use), "bound": True}
    except Exception:
        return None


def parse_cpulist(text: str) -> List[int]:
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11] (the kernel's cpulist format)."""
    out: List[int] = []
    for part in text.strip().split(","):
        if not part:
            continue
        if "-" in part:
            a, b = part.split("-")
            out.extend(range(int(a), int(b) + 1))
        else:
            out.append(int(part))
    return out
