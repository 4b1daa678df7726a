#!/usr/bin/env python
"""Per-shape durations of tools/gemm_small_probe.py from the rocprofv3 rocpd database (argv[1])."""
import sqlite3, sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
REPS = 20
SHAPES = [(1024, 256), (1024, 512), (1024, 1024), (1024, 2048), (1024, 4096), (2048, 1024), (512, 1024), (4096, 1024)]
if os.environ.get("PROBE_SHAPES"):   # "NxK,NxK,..."
    SHAPES = [tuple(int(v) for v in t.split("x")) for t in os.environ["PROBE_SHAPES"].split(",")]
M = int(os.environ.get("PROBE_M", "1536"))
db = sqlite3.connect(sys.argv[1])
rows = list(db.cursor().execute("select name, grid_x, grid_y, grid_z, end-start from kernels where name like '%gemm_split_glds_kernel%' order by start"))
assert len(rows) == REPS * len(SHAPES), (len(rows), REPS * len(SHAPES))
for i, (N, K) in enumerate(SHAPES):
    r = rows[i * REPS:(i + 1) * REPS]
    d = sorted(x[4] for x in r[5:])
    med = d[len(d) // 2] / 1e3
    print(f"M={M} N={N} K={K}: grid=({r[0][1]},{r[0][2]},{r[0][3]}) {r[0][0][28:60]} median {med:.1f} us min {d[0]/1e3:.1f} us  -> {2.0*M*N*K/med/1e6:.0f} TF fp32-equiv, {med*1e3/(K/32):.0f} ns per k-tile")
