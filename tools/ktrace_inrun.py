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


if __name__ == "__main__":
    print(measure("ar_attn_fused_kernel", 16, int(sys.argv[1]) if len(sys.argv) > 1 else 2100, "f16", "f16"))
