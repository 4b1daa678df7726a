#!/bin/bash
# Route M step (tools/ab_ln_fold.py BATCH) under rocprofv3 --kernel-trace: per-kernel totals.  usage: bash tools/profile_m.sh <tag> <batch> [env assignments...]
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; B=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_m
env "$@" rocprofv3 --kernel-trace -d $R/gpurun_out/prof_m -o m -- python $R/tools/ab_ln_fold.py $B 2 > $R/gpurun_out/${TAG}.log 2>&1
DB=$(find $R/gpurun_out/prof_m -name "*.db" | head -1)
python $R/tools/rocpd_kernel_stats.py $DB > $R/gpurun_out/${TAG}_kernel_stats.csv
cp $DB $R/gpurun_out/${TAG}.db
tail -1 $R/gpurun_out/${TAG}.log; head -14 $R/gpurun_out/${TAG}_kernel_stats.csv | cut -c1-150
rm -rf $R/gpurun_out/prof_m
