"""The scene-parallel exchange on real hardware as far as a 1-GPU box allows: a process group with backend "nccl" (= RCCL on ROCm) of ONE rank, the same calls bench.py and
bevgen_amd.parallel make at N > 1 - barrier, all_gather of the per-rank timings, the gather of uint8 pixels and of token ids (forced through the collective:
$BEVGEN_FORCE_COLLECTIVE) - in a child process (a process group is per-process state).  RCCL refuses two ranks on one device, so the multi-rank control flow stays with the
gloo tests (tests/test_parallel_cpu.py, tests/test_bench_cpu.py); what this adds is that the RCCL call path itself has run on an MI355X."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

CHILD = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["BEVGEN_REPO"])
from bevgen_amd.parallel import gather_scenes, gather_token_ids
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
dist.barrier()
torch.cuda.synchronize()
t = torch.tensor([1.5, 2.5], dtype=torch.float64, device="cuda")
out = [torch.empty_like(t)]
dist.all_gather(out, t)
assert torch.equal(out[0], t)
g = torch.Generator(device="cuda").manual_seed(7)
px = torch.randint(0, 256, (2, 6, 3, 32, 32), dtype=torch.uint8, device="cuda", generator=g)
got = gather_scenes(px, dist)
assert got is not None and got.dtype == torch.uint8 and torch.equal(got, px)
fl = torch.rand((2, 6, 3, 32, 32), device="cuda", generator=g)
got = gather_scenes(fl, dist)
assert torch.equal(got, (fl * 255.0).round().clamp(0, 255).to(torch.uint8))
ids = torch.randint(0, 1024, (12, 16, 16), dtype=torch.int64, device="cuda", generator=g)
got = gather_token_ids(ids, dist)
assert got.dtype == torch.int64 and torch.equal(got, ids)
dist.barrier()
dist.destroy_process_group()
print("rccl world-1 ok")
"""


def test_rccl_collectives_of_the_scene_parallel_path_with_one_rank():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", BEVGEN_FORCE_COLLECTIVE="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0", BEVGEN_REPO=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "rccl world-1 ok" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
