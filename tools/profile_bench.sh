#!/bin/bash
# Round artefacts: default bench line + rocprofv3 kernel stats of the same command (shortened), copied into gpurun_out/ for profiles/.
# usage on the GPU box: bash tools/profile_bench.sh <tag>
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-rXX}
cd $R && python bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err
tail -c 600 gpurun_out/${TAG}_bench_default.err
cp gpurun_out/bench_detail_n1.json gpurun_out/${TAG}_bench_detail_n1.json   # (the shortened run below rewrites the detail file)
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_bench
BEVGEN_BENCH_NO_PMC=1 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_bench -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra-legs --no-decode-leg --no-exact-leg > $R/gpurun_out/${TAG}_bench_under_rocprof.json 2> $R/gpurun_out/prof_bench.err
DB=$(find $R/gpurun_out/prof_bench -name "*.db" | head -1)
python $R/tools/rocpd_kernel_stats.py $DB > $R/gpurun_out/${TAG}_bench_kernel_stats.csv
rm -rf $R/gpurun_out/prof_bench
head -c 1500 $R/gpurun_out/${TAG}_bench_default.json; echo; head -12 $R/gpurun_out/${TAG}_bench_kernel_stats.csv | cut -c1-160
