#!/bin/bash
# usage on the GPU box: bash tools/gemm_small.sh <tag>   (honours BEVGEN_GEMM_STAGES, PROBE_M)
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-x}
cd /tmp && export TMPDIR=/tmp
export BEVGEN_GEMM_ROWSPLIT=${BEVGEN_GEMM_ROWSPLIT:-0}   # one launch per call: the report counts launches per shape
rm -rf $R/gpurun_out/prof_gs
rocprofv3 --kernel-trace -d $R/gpurun_out/prof_gs -o gs -- python $R/tools/gemm_small_probe.py > /dev/null 2> $R/gpurun_out/prof_gs.err
DB=$(find $R/gpurun_out/prof_gs -name "*.db" | head -1)
python $R/tools/gemm_small_report.py $DB | tee $R/gpurun_out/gemm_small_$TAG.txt
rm -rf $R/gpurun_out/prof_gs
