#!/usr/bin/env python
"""Consecutive kernels of a rocprofv3 (rocpd sqlite) trace with grid and duration, averaged over the repetitions of a periodic pattern: usage rocpd_sequence.py db start_frac period
prints, for positions 0..period-1 of the launch sequence beginning at the first `layernorm`/`ln_stats` after start_frac of the trace, the mean duration over 40 periods."""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1]); frac = float(sys.argv[2]); period = int(sys.argv[3])
rows = list(db.cursor().execute("select name, grid_x, grid_y, start, end from kernels order by start"))
i0 = int(len(rows) * frac)
acc = [[0.0, 0, ""] for _ in range(period)]
for rep in range(40):
    for p in range(period):
        n, gx, gy, s, e = rows[i0 + rep * period + p]
        short = re.sub(r"\(.*", "", n).replace("void bevgen::", "").replace("bevgen::", "")[:48]
        acc[p][0] += (e - s) / 1e3; acc[p][1] += 1; acc[p][2] = f"{short} ({gx},{gy})"
for p in range(period):
    print(f"{p:3d} {acc[p][2]:75s} {acc[p][0]/acc[p][1]:8.1f} us")
