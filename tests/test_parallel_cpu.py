"""Scene-parallel sharding + the final gather, world_size 2 on the gloo backend (the GPU path uses the same code over RCCL)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from bevgen_amd.parallel import GatherMismatch, gather_scenes, gather_token_ids, parse_cpulist, payload_checksum, shard_range, to_uint8


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _scene_pixels(scene_idx: int, C=2, H=4, W=4):
    g = torch.Generator().manual_seed(1000 + scene_idx)
    return torch.rand(C, 3, H, W, generator=g)


def _worker(rank, world, port, total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    b, e = shard_range(total, rank, world)
    px = torch.stack([_scene_pixels(i) for i in range(b, e)])
    ids = torch.stack([torch.full((2, 5), i, dtype=torch.int64) for i in range(b, e)])
    out = gather_scenes(px, dist)
    out_ids = gather_token_ids(ids, dist)
    if rank == 0:
        q.put((out, out_ids))
    else:
        assert out is None and out_ids is None
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_partitions_everything():
    for total in (1, 7, 16, 64, 129):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1


def test_gather_world2_gloo():
    world, total = 2, 6
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    out, out_ids = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    want = to_uint8(torch.stack([_scene_pixels(i) for i in range(total)]))
    assert out.dtype == torch.uint8 and torch.equal(out, want)
    assert torch.equal(out_ids[:, 0, 0], torch.arange(total))


def test_single_process_passthrough():
    px = torch.rand(3, 2, 3, 4, 4)
    assert torch.equal(gather_scenes(px, None), to_uint8(px))


def _corrupting_worker(rank, world, port, q):
    """Rank 1's payload is altered AFTER its checksum was taken (stand-in for a block damaged on the way): rank 0 must refuse the gather."""
    import bevgen_amd.parallel as P

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    px = to_uint8(torch.stack([_scene_pixels(10 + rank)]))
    if rank == 1:
        real, intact = P.payload_checksum, px
        P.payload_checksum = lambda t: real(intact)      # the checksum of the intact block ...
        px = px.clone()
        px.view(-1)[7] ^= 0x10                            # ... travels beside a block with one flipped bit
    try:
        gather_scenes(px, dist)
        q.put((rank, "ok"))
    except GatherMismatch as e:
        q.put((rank, str(e)))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_checksum_catches_a_damaged_block():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_corrupting_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert "ranks [1]" in res[0], res
    assert res[1] == "ok"


def test_payload_checksum_and_cpulist():
    a = torch.arange(24, dtype=torch.uint8).reshape(2, 3, 4)
    assert not torch.equal(payload_checksum(a), payload_checksum(a.transpose(1, 2).contiguous()))   # same multiset, different order
    assert not torch.equal(payload_checksum(a), payload_checksum(torch.zeros_like(a)))
    assert torch.equal(payload_checksum(a), payload_checksum(a.clone()))
    assert parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    assert parse_cpulist("5") == [5] and parse_cpulist("") == []
