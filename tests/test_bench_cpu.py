"""bench.py's multi-rank control flow, executed on CPU before the first real multi-GPU run: BEVGEN_BENCH_DRYRUN=1 swaps the library context for a shape-only stub
and RCCL for gloo; rank sharding, barrier + max-over-ranks timing, the uint8 gather to rank 0, the strong-scaling leg and rank-0-only reporting are the real code."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(cmd, extra_env=None):
    env = dict(os.environ, BEVGEN_BENCH_DRYRUN="1", MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    env.update(extra_env or {})
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)


def test_bench_world_2_control_flow_under_gloo():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "4"]
    r = _run(cmd)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, f"exactly one JSON line (rank 0 only), got {len(lines)}"
    d = json.loads(lines[0])
    assert d["dry_run"] is True and d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert d["config"]["global_batch"] == 8 and "x2" in d["config"]["parallelism"]
    assert abs(d["value"] - 2 * 4 * 3 / (d["ms_per_step"] * 3e-3)) / d["value"] < 1e-6     # whole-job scenes / max-over-ranks time
    s = d["strong_scaling"]
    assert s["scaling"] == "strong" and s["global_batch"] == 16 and s["scenes_per_gpu"] == 8 and s["value"] > 0
    for k in ("roofline", "cpu_baseline", "ms_per_decode_step"):     # single-GPU legs stay off in a multi-rank run (cpu_baseline: rank 0 at N = 1 only)
        assert k == "roofline" or k not in d


def test_bench_self_launch_and_world_size_mismatch():
    # without a launcher: --gpus 2 re-executes itself under torch.distributed.run with 2 ranks
    r = _run([sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2"])
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 4
    # a launcher that started a different number of ranks than --gpus says: refuse loudly
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           "bench.py", "--gpus", "4", "--steps", "2", "--warmup", "1"]
    r = _run(cmd)
    assert r.returncode != 0 and "--gpus 4 but the launcher started 2" in (r.stderr + r.stdout)


def test_bench_single_rank_dry_run_line_shape():
    r = _run([sys.executable, "bench.py", "--steps", "2", "--warmup", "1", "--batch", "2"])
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 1 and d["metric"].startswith("multi-view scenes/sec") and d["unit"] == "scenes/s" and d["vs_baseline"] is None and "strong_scaling" not in d
