#!/bin/bash
# Route A decode under rocprofv3 --kernel-trace; writes gpurun_out/<tag>_decode_kernel_stats.csv
# usage on the GPU box: bash tools/profile_decode.sh <tag> [decode_probe args...]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-rXX}; shift
rm -rf $R/gpurun_out/dec_prof
rocprofv3 --kernel-trace -d $R/gpurun_out/dec_prof -o dec -- python $R/tools/decode_probe.py "$@" > $R/gpurun_out/${TAG}_decode_probe.log 2>&1
DB=$(find $R/gpurun_out/dec_prof -name "*.db" | head -1)
python $R/tools/rocpd_kernel_stats.py $DB > $R/gpurun_out/${TAG}_decode_kernel_stats.csv
python $R/tools/rocpd_gaps.py $DB 5100 > $R/gpurun_out/${TAG}_decode_gaps.txt   # the last ~100 steps: kernel time + idle gaps of the replayed chain
tail -3 $R/gpurun_out/${TAG}_decode_probe.log
head -14 $R/gpurun_out/${TAG}_decode_kernel_stats.csv | cut -c1-200
rm -rf $R/gpurun_out/dec_prof
