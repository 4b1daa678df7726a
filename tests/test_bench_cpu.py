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
    assert d["rccl_world_size"] == 2 and len(d["per_rank_ms_per_step"]) == 2 and max(d["per_rank_ms_per_step"]) <= d["ms_per_step"] * 1.0001   # per-rank K-step times <= the job's
    legs = d["legs"]
    assert list(d)[-2:] == ["legs", "roofline_decode_attention"] or list(d)[-1] == "legs", "the flat per-leg scalars (and the decode-attention roofline) come LAST in the line: they must survive a tail cut"
    s = legs["strong_scaling"]
    assert s["global_batch"] == 16 and s["scenes_per_s"] > 0 and len(s["per_rank_ms_per_step"]) == 2
    c5 = legs["config5"]       # BASELINE configs[4] across the ranks: 64 sequences per GPU, token ids gathered to rank 0
    assert c5["sequences_per_gpu"] == 64 and c5["sequences_per_s"] > 0 and len(c5["per_rank_ms_per_decode_step"]) == 2
    assert abs(c5["sequences_per_s"] - 2 * 64 / (c5["ms_per_decode_step"] * c5["decode_steps"] * 1e-3)) / c5["sequences_per_s"] < 1e-2
    assert "cpu_baseline" not in d and "decode_config4_B16" not in legs      # single-GPU legs stay off in a multi-rank run (cpu_baseline: rank 0 at N = 1 only)
    assert "roofline" in d and len(lines[0]) < 4000, "the printed line stays short; the per-leg objects live in the detail file"
    detail = json.load(open(os.path.join(ROOT, d["detail_file"])))
    assert detail["config5"]["gathered_token_ids"][0] == 128 and detail["strong_scaling"]["scaling"] == "strong"


def test_bench_world_8_control_flow_under_gloo():
    """The real rank count of BASELINE configs[2] / configs[4]: 8 ranks (gloo, stub context): weak leg 16 scenes per rank, the strong leg 16 / 8 = 2 scenes per rank,
    config 5 with 64 sequences per rank gathered to rank 0 (512 rows)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           "bench.py", "--gpus", "8", "--steps", "2", "--warmup", "1"]
    r = _run(cmd, {"OMP_NUM_THREADS": "1"})
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["dry_run"] is True and d["n_gpus"] == 8 and d["rccl_world_size"] == 8 and d["config"]["global_batch"] == 128 and len(d["per_rank_ms_per_step"]) == 8
    assert abs(d["value"] - 8 * 16 * 2 / (d["ms_per_step"] * 2e-3)) / d["value"] < 1e-6
    s = d["legs"]["strong_scaling"]
    assert s["global_batch"] == 16 and len(s["per_rank_ms_per_step"]) == 8
    c5 = d["legs"]["config5"]
    assert c5["sequences_per_gpu"] == 64 and len(c5["per_rank_ms_per_decode_step"]) == 8
    detail = json.load(open(os.path.join(ROOT, d["detail_file"])))
    assert detail["strong_scaling"]["scenes_per_gpu"] == 2 and detail["config5"]["gathered_token_ids"][0] == 512


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
    assert d["n_gpus"] == 1 and d["metric"].startswith("multi-view scenes/sec") and d["unit"] == "scenes/s" and d["vs_baseline"] is None
    assert "strong_scaling" not in d["legs"] and "config5" not in d["legs"] and d["rccl_world_size"] == 1 and d["roofline"]["traffic"] is None
    # the HBM roofline of the north star is a first-class sibling of `roofline` (values only on a GPU), and no leg prints a "fraction of peak" that a window artefact
    # can push above 1
    ra = d["roofline_decode_attention"]
    assert ra["bound"] == "hbm" and ra["unit"] == "GB/s" and {"kernel", "launches", "avg_us", "frac", "traffic", "achieved", "peak"} <= set(ra)
    assert "attn_phase_frac" not in json.dumps(d)


def test_bench_single_gpu_predictions_of_the_scaling_legs():
    """At the default batch (16) the N = 1 line states what one GPU does at the per-GPU load of the 2 / 4 / 8-GPU strong-scaling legs (batch 8 / 4 / 2)."""
    r = _run([sys.executable, "bench.py", "--steps", "2", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    p = d["legs"]["scaling_prediction_from_one_gpu"]
    assert set(p["strong_16_scenes"]) == {"2", "4", "8"} and set(p["per_gpu_scenes_per_s_at_batch"]) == {"8", "4", "2"} and set(p["weak_16_per_gpu"]) == {"2", "4", "8"}
    assert abs(p["weak_16_per_gpu"]["8"] - 8 * d["value"]) / d["value"] < 1e-2
