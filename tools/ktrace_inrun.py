#!/usr/bin/env python
"""Average kernel duration of the fused decode-attention kernel on the PRODUCT path (hipGraph replay of the decode step), measured by a `rocprofv3 --kernel-trace`
child pass over tools/decode_probe.py - bench.py calls measure() for its fp16-cache + fp16-weights decode leg, outside every timed region (rank 0, N = 1).
The per-launch HIP-event pairs of the bench's eager profiling pass time the kernel PLUS the dispatch latency of its launch (~2 us on a 26 us kernel); the trace times the kernel."""
import glob
import os
import shutil
import sqlite3
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def measure(kernel_substr, batch, steps, kv, weights, timeout=300):
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {"error": "rocprofv3 not found"}
    d = tempfile.mkdtemp(prefix="bevgen_kt_", dir="/tmp")
    try:
        cmd = [exe, "--kernel-trace", "-d", d, "-o", "kt", "--", sys.executable, os.path.join(ROOT, "tools", "decode_probe.py"), str(batch), str(steps), "fused", kv, "1", weights]
        r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=timeout)
        dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
        if r.returncode != 0 or not dbs:
            return {"error": f"rocprofv3 --kernel-trace: rc {r.returncode}, {len(dbs)} db; {r.stderr[-300:]}"}
        db = sqlite3.connect(dbs[0])
        cur = db.cursor()
        cols = [c[1] for c in cur.execute("pragma table_info('kernels')")]
        name_col = "name" if "name" in cols else cols[0]
        rows = list(cur.execute(f"select {name_col}, count(*), avg(end-start) from kernels where {name_col} like ? group by {name_col} order by 2 desc", (f"%{kernel_substr}%",)))
        if not rows:
            return {"error": f"no kernel containing '{kernel_substr}' in the trace"}
        name, calls, avg_ns = rows[0]
        return {"kernel": " ".join(name.split())[:90], "launches": int(calls), "avg_us": avg_ns / 1e3,
                "source": f"in-run rocprofv3 --kernel-trace child pass over tools/decode_probe.py {batch} {steps} fused {kv} 1 {weights} (hipGraph replay of the decode step; its 8 warm-up steps included)"}
    except subprocess.TimeoutExpired:
        return {"error": f"rocprofv3 --kernel-trace: timeout after {timeout} s"}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def measure_traffic(kernel_substr, batch, steps, kv, weights, timeout=300):
    """HBM traffic per launch of the decode-attention kernel, measured in this run: two `rocprofv3 --kernel-trace --pmc` child passes (FETCH_SIZE, WRITE_SIZE: separate
    passes, kernel trace only - MI355X_MICROARCH.md, HBM / rocprofv3 sections) over tools/decode_probe.py for `steps` decode steps (contexts 257 ... 256 + steps, plus the
    probe's 8 warm-up steps at contexts 257 ... 264), with the guide's gfx950 correction (read = 2 x FETCH_SIZE KiB, write = WRITE_SIZE KiB), against the algorithmic bytes
    of the same launches: K and V rows of the context once per (sequence, head) + this layer's q/k/v weight matrix once."""
    import csv
    sys.path.insert(0, ROOT)
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {"error": "rocprofv3 not found"}
    got = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="bevgen_kpmc_", dir="/tmp")
        try:
            cmd = [exe, "--kernel-trace", "--pmc", counter, "-d", d, "-o", "pmc", "--output-format", "csv", "--", sys.executable, os.path.join(ROOT, "tools", "decode_probe.py"),
                   str(batch), str(steps), "fused", kv, "1", weights]
            r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=timeout)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return {"error": f"rocprofv3 --pmc {counter}: rc {r.returncode}, {len(files)} csv; {r.stderr[-300:]}"}
            vals = [float(row["Counter_Value"]) for row in csv.DictReader(open(files[0])) if row.get("Counter_Name") == counter and kernel_substr in row.get("Kernel_Name", "")]
            if not vals:
                return {"error": f"no {counter} rows for a kernel containing '{kernel_substr}'"}
            got[counter] = (sum(vals) / len(vals), len(vals))
        except subprocess.TimeoutExpired:
            return {"error": f"rocprofv3 --pmc {counter}: timeout after {timeout} s"}
        finally:
            shutil.rmtree(d, ignore_errors=True)
    from bevgen_amd import presets
    cfg = presets.config4()
    K, H, D = cfg.num_cond_tokens, cfg.num_heads, cfg.num_embed
    ctxs = [K + s + 1 for s in range(8)] + [K + s + 1 for s in range(steps)]      # decode_probe: 8 warm-up steps, then `steps`
    mean_n = sum(ctxs) / len(ctxs)
    kvb, wb = (2 if kv == "f16" else 4), (2 if weights == "f16" else 4)
    alg = 2.0 * H * 64 * kvb * batch * mean_n + 3.0 * D * D * wb
    rd, wr = 2.0 * got["FETCH_SIZE"][0] * 1024.0, got["WRITE_SIZE"][0] * 1024.0
    return {"bytes_per_launch": rd + wr, "read_bytes_corrected": rd, "write_bytes": wr, "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": (rd + wr) / alg,
            "launches": got["FETCH_SIZE"][1], "context": f"mean over {len(ctxs)} steps x {cfg.num_layers} layers, mean context {mean_n:.0f} keys, B={batch}, {kv} K/V, {weights} weights",
            "source": "in-run rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE child passes over tools/decode_probe.py; read = 2 x FETCH_SIZE KiB (gfx950 correction), write = WRITE_SIZE KiB"}


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1].startswith("traffic"):
        print(measure_traffic("ar_attn_fused_kernel", 16, int(sys.argv[1][7:] or 16), "f16", "f16"))
        sys.exit(0)
    print(measure("ar_attn_fused_kernel", 16, int(sys.argv[1]) if len(sys.argv) > 1 else 2100, "f16", "f16"))
