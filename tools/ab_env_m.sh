#!/bin/bash
# Same-box A/B of the Route M headline step over an environment switch:  bash tools/ab_env_m.sh VAR "v1 v2 ..." [bench args]
R=${GRAFT_REPO_ROOT:-/root/repo}
export BEVGEN_BENCH_NO_PMC=1
VAR=$1; VALS=$2; shift 2
for i in 1 2; do
for v in $VALS; do
  export $VAR=$v
  python $R/bench.py --steps 3 --warmup 1 --no-decode-leg --no-extra-legs --no-cpu-baseline --no-exact-leg "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$VAR=$v', 'scenes/s', round(d['value'],3), 'ms/step', round(d['ms_per_step'],1), 'it', round(d['legs']['ms_per_maskgit_iteration'],2), {k: round(x,3) for k,x in d['legs']['kernel_time_share'].items()}, {k: round(x,1) for k,x in d['legs']['kernel_tflops'].items()})"
done; done
