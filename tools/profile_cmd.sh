#!/bin/bash
# rocprofv3 --kernel-trace of an arbitrary python command; prints the top kernel stats.  usage: bash tools/profile_cmd.sh <tag> <python args...>
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
rm -rf $R/gpurun_out/cmd_prof
rocprofv3 --kernel-trace -d $R/gpurun_out/cmd_prof -o t -- python "$@" > $R/gpurun_out/${TAG}.log 2>&1
DB=$(find $R/gpurun_out/cmd_prof -name "*.db" | head -1)
python $R/tools/rocpd_kernel_stats.py $DB > $R/gpurun_out/${TAG}_stats.csv
echo "== $TAG"; head -4 $R/gpurun_out/${TAG}_stats.csv | cut -c1-180
rm -rf $R/gpurun_out/cmd_prof
