#!/bin/bash
# Same-box A/B of the Route M headline step: this tree's library vs another build (.ab/lib<name>.so), alternating.  usage on the GPU box: bash tools/ab_lib_m.sh [other=prev] [bench args]
R=${GRAFT_REPO_ROOT:-/root/repo}
export BEVGEN_BENCH_NO_PMC=1
OTHER=${1:-prev}; shift
for i in 1 2 3; do
for lib in ${ORDER:-new $OTHER}; do
  if [ $lib = new ]; then unset BEVGEN_LIB_PATH; else export BEVGEN_LIB_PATH=$R/.ab/lib$lib.so; fi
  python $R/bench.py --steps 3 --warmup 1 --no-decode-leg --no-extra-legs --no-cpu-baseline --no-exact-leg "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', 'scenes/s', round(d['value'],3), 'ms/step', round(d['ms_per_step'],1), {k: round(x,3) for k,x in d['legs']['kernel_time_share'].items()}, {k: round(x,1) for k,x in d['legs']['kernel_tflops'].items()})"
done; done
