#!/bin/bash
# one Route M bench step under rocprofv3 --kernel-trace; writes gpurun_out/step_stats.csv  (usage on the GPU box: bash tools/profile_step.sh [extra bench args])
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/step_prof
rocprofv3 --kernel-trace -d $R/gpurun_out/step_prof -o step -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-decode-leg --no-exact-leg "$@" > $R/gpurun_out/step_bench.log 2>&1
DB=$(find $R/gpurun_out/step_prof -name "*.db" | head -1)
python $R/tools/rocpd_kernel_stats.py $DB > $R/gpurun_out/step_stats.csv
tail -1 $R/gpurun_out/step_bench.log | cut -c1-300
head -25 $R/gpurun_out/step_stats.csv | cut -c1-200
rm -rf $R/gpurun_out/step_prof
