#!/bin/bash
# Same-box matrix of the small-batch GEMM switches on the Route M step: small-problem block of the 128-row tile (BEVGEN_GEMM_STAGES 2 | 8) x split-K (BEVGEN_KSPLIT).
# usage on the GPU box: bash tools/ab_b1_gemm.sh [batch=1] ["stages list"] ["ksplit list"]
R=${GRAFT_REPO_ROOT:-/root/repo}
B=${1:-1}; SL=${2:-"2 8"}; KL=${3:-"1 2 3"}
for i in 1 2; do
for st in $SL; do for ks in $KL; do
  export BEVGEN_GEMM_STAGES=$st BEVGEN_KSPLIT=$ks
  python $R/bench.py --batch $B --steps 3 --warmup 1 --no-decode-leg --no-extra-legs --no-cpu-baseline --no-exact-leg 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=$B stages=$st ksplit=$ks', 'ms/scene', round(d['ms_per_step']/$B,1), 'it', round(d['ms_per_maskgit_iteration'],2), {k: round(x,3) for k,x in d['kernel_time_share'].items()})"
done; done; done
