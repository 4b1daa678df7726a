import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1])
rows = list(db.cursor().execute("select name, grid_x, grid_y, start, end from kernels order by start"))
i0 = int(len(rows) * 0.6)
prev_end = None
for n, gx, gy, s, e in rows[i0:i0 + 48]:
    short = re.sub(r"\(.*", "", n).replace("void bevgen::", "").replace("bevgen::", "")[:60]
    gap = (s - prev_end) / 1e3 if prev_end else 0
    print(f"{short:62s} ({gx},{gy}) {(e-s)/1e3:7.1f} us gap {gap:6.1f}")
    prev_end = e
