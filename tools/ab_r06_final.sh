#!/bin/bash
# round-6 closing A/Bs on one box: LayerNorm fold level 1 vs 2 after the row-major epilogues (16 / 2 / 1 scenes), then the power trace of the current GEMM
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
ROUNDS=2 bash tools/ab.sh m env BEVGEN_LN_FOLD=1,2 > $O/ab_lnfold_b16.txt 2>&1
: > $O/ab_lnfold_small.txt
for i in 1 2; do for b in 1 2; do for f in 1 2; do
  BEVGEN_LN_FOLD=$f python tools/ab_ln_fold.py $b 5 2>/dev/null | tail -1 >> $O/ab_lnfold_small.txt
done; done; done
bash tools/power/power_trace.sh r06 > $O/r06_power_summary.txt 2>&1
cat $O/ab_lnfold_b16.txt $O/ab_lnfold_small.txt; tail -12 $O/r06_power_summary.txt
