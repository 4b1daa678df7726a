#!/usr/bin/env python
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace into the per-kernel stats table that `--stats` prints in CSV mode:
name, calls, total ns, average ns, min, max, percentage.   usage: rocpd_kernel_stats.py results.db > stats.csv"""
import re
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [c[1] for c in cur.execute("pragma table_info('kernels')")]
    name_col = "name" if "name" in cols else cols[0]
    rows = list(cur.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by {name_col} order by 3 desc"))
    total = sum(r[2] for r in rows) or 1
    print("Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage")
    for name, calls, tot, avg, mn, mx in rows:
        short = re.sub(r"\s+", " ", name)
        print(f"\"{short}\",{calls},{tot},{avg:.1f},{mn},{mx},{100.0 * tot / total:.3f}")


if __name__ == "__main__":
    main(sys.argv[1])
