#!/bin/bash
# Route M at a small batch under rocprofv3 --kernel-trace: per-(kernel, grid) durations + the wall time of the same run.
# usage on the GPU box: bash tools/profile_b1.sh <tag> [batch=1]
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-rXX}; B=${2:-1}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_b1
BEVGEN_BENCH_NO_PMC=1 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_b1 -o b1 -- python $R/bench.py --batch $B --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs --no-decode-leg --no-exact-leg > $R/gpurun_out/${TAG}_b${B}_under_rocprof.json 2> $R/gpurun_out/prof_b1.err
DB=$(find $R/gpurun_out/prof_b1 -name "*.db" | head -1)
python $R/tools/rocpd_by_grid.py $DB 40 > $R/gpurun_out/${TAG}_b${B}_by_grid.txt
python $R/tools/rocpd_kernel_stats.py $DB > $R/gpurun_out/${TAG}_b${B}_kernel_stats.csv
python - <<PY
import sqlite3
db = sqlite3.connect("$DB")
c = db.cursor()
n, tot, t0, t1 = list(c.execute("select count(*), sum(end-start), min(start), max(end) from kernels"))[0]
print(f"kernels {n}, busy {tot/1e6:.1f} ms, span {(t1-t0)/1e6:.1f} ms")
# idle time BETWEEN consecutive kernels (gaps below 100 us: back-to-back launches of one forward, not host work between steps)
rows = list(c.execute("select start, end from kernels order by start"))
gaps = [b[0] - a[1] for a, b in zip(rows, rows[1:])]
small = [g for g in gaps if 0 <= g < 100000]
import statistics
print(f"gaps < 100 us between consecutive kernels: {len(small)} of {len(gaps)}, total {sum(small)/1e6:.1f} ms, median {statistics.median(small)/1e3:.2f} us, mean {sum(small)/len(small)/1e3:.2f} us; overlapping starts {sum(1 for g in gaps if g < 0)}")
PY
rm -rf $R/gpurun_out/prof_b1
cat $R/gpurun_out/${TAG}_b${B}_by_grid.txt | head -40
