#!/bin/bash
# Route M at a small batch under rocprofv3 --kernel-trace: per-(kernel, grid) durations + the wall time of the same run.
# usage on the GPU box: bash tools/profile_b1.sh <tag> [batch=1]
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-rXX}; B=${2:-1}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_b1
rocprofv3 --kernel-trace -d $R/gpurun_out/prof_b1 -o b1 -- python $R/bench.py --batch $B --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs --no-decode-leg --no-exact-leg > $R/gpurun_out/${TAG}_b${B}_under_rocprof.json 2> $R/gpurun_out/prof_b1.err
DB=$(find $R/gpurun_out/prof_b1 -name "*.db" | head -1)
python $R/tools/rocpd_by_grid.py $DB 40 > $R/gpurun_out/${TAG}_b${B}_by_grid.txt
python $R/tools/rocpd_kernel_stats.py $DB > $R/gpurun_out/${TAG}_b${B}_kernel_stats.csv
python - <<PY
import sqlite3
db = sqlite3.connect("$DB")
c = db.cursor()
n, tot, t0, t1 = list(c.execute("select count(*), sum(end-start), min(start), max(end) from kernels"))[0]
print(f"kernels {n}, busy {tot/1e6:.1f} ms, span {(t1-t0)/1e6:.1f} ms")
PY
rm -rf $R/gpurun_out/prof_b1
cat $R/gpurun_out/${TAG}_b${B}_by_grid.txt | head -40
