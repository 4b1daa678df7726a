#!/usr/bin/env python
"""Summarise sampler.py traces: per card, median / max power and median / min clock over the busy part of each trace (samples above the trace's idle level)."""
import csv, sys

for path in sys.argv[1:]:
    lines = [l for l in open(path) if not l.startswith("#")]
    rows = list(csv.reader(lines))
    head, rows = rows[0], [[float(v) for v in r] for r in rows[1:] if len(r) == len(rows[0])]
    ncard = (len(head) - 1) // 2
    print(path.split("/")[-1], f"{len(rows)} samples")
    for c in range(ncard):
        p = [r[1 + 2 * c] for r in rows]; s = [r[2 + 2 * c] for r in rows]
        if not p: continue
        lo, hi = min(p), max(p)
        busy = [i for i in range(len(p)) if p[i] > lo + 0.5 * (hi - lo)] if hi - lo > 50 else list(range(len(p)))
        pb = sorted(p[i] for i in busy); sb = sorted(s[i] for i in busy)
        print(f"  card {c}: power W min {lo:.0f} max {hi:.0f}; over the {len(busy)} busy samples: power median {pb[len(pb)//2]:.0f}, sclk median {sb[len(sb)//2]:.0f} MHz (min {sb[0]:.0f}, max {sb[-1]:.0f})")
