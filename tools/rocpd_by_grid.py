#!/usr/bin/env python
"""Per-(kernel, grid) durations from a rocprofv3 rocpd database: name, grid_x, grid_y, calls, avg ns, min ns, total ms."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
q = ("select name, grid_x, grid_y, count(*), avg(end-start), min(end-start), sum(end-start)/1e6 from kernels "
     "group by name, grid_x, grid_y order by sum(end-start) desc limit %d" % (int(sys.argv[2]) if len(sys.argv) > 2 else 16))
for r in db.cursor().execute(q):
    print(f"{r[0][:58]:58s} grid=({r[1]},{r[2]}) calls={r[3]} avg={r[4]/1e3:.1f}us min={r[5]/1e3:.1f}us total={r[6]:.1f}ms")
