#!/usr/bin/env python
"""Where a chain of dependent kernels spends its time: from a rocprofv3 (rocpd sqlite) kernel trace, the LAST `count` kernels (e.g. the tail of a decode: whole replayed
steps) as busy time per kernel name + the idle gaps between consecutive kernels.   usage: rocpd_gaps.py results.db [count=7500]"""
import re
import sqlite3
import sys
from collections import defaultdict


def main(path, count=7500):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [c[1] for c in cur.execute("pragma table_info('kernels')")]
    name_col = "name" if "name" in cols else cols[0]
    rows = list(cur.execute(f"select {name_col}, start, end from kernels order by start"))[-count:]
    busy, calls, gaps, gap_after = defaultdict(int), defaultdict(int), 0, defaultdict(int)
    for i, (name, s, e) in enumerate(rows):
        short = re.sub(r"\s+", " ", name).replace("void bevgen::", "").split("(")[0][:70]
        busy[short] += e - s
        calls[short] += 1
        if i + 1 < len(rows):
            g = max(0, rows[i + 1][1] - e)
            gaps += g
            gap_after[short] += g
    span = rows[-1][2] - rows[0][1]
    print(f"{len(rows)} kernels, span {span / 1e3:.1f} us, busy {sum(busy.values()) / 1e3:.1f} us, gaps {gaps / 1e3:.1f} us ({100.0 * gaps / span:.1f} %)")
    for k in sorted(busy, key=busy.get, reverse=True):
        print(f"  {k:<72} calls {calls[k]:>6}  avg {busy[k] / calls[k] / 1e3:7.2f} us  busy {100.0 * busy[k] / span:5.1f} %  avg gap after {gap_after[k] / calls[k] / 1e3:5.2f} us")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 7500)
