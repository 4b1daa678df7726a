#!/bin/bash
# HBM traffic of the rooflined kernels from the PMC counters (separate --pmc passes, kernel trace only; MI355X guide, HBM section).
# usage on the GPU box: bash tools/pmc_collect.sh <tag>   -> gpurun_out/<tag>_pmc_fetch_size.csv, _pmc_write_size.csv, _pmc_hbm_traffic.json
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-rXX}
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_$c
  timeout 400 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/pmc_$c -o pmc --output-format csv -- python $R/tools/pmc_probe.py ${PMC_DECODE_STEPS:-40} > $R/gpurun_out/pmc_$c.log 2>&1
  cp $(find $R/gpurun_out/pmc_$c -name "*counter_collection.csv" | head -1) $R/gpurun_out/${TAG}_pmc_$(echo $c | tr A-Z a-z).csv
  rm -rf $R/gpurun_out/pmc_$c
done
python $R/tools/pmc_summarise.py $R/gpurun_out/${TAG}_pmc_fetch_size.csv $R/gpurun_out/${TAG}_pmc_write_size.csv > $R/gpurun_out/${TAG}_pmc_hbm_traffic.json
cat $R/gpurun_out/${TAG}_pmc_hbm_traffic.json
