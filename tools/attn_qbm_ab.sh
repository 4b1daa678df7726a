#!/bin/bash
# same-box A/B of the workgroup order of attention_split_kernel ($BEVGEN_ATTN_QB_MAJOR) over batch sizes; flat queue off
cd ${GRAFT_REPO_ROOT:-/root/repo}
export BEVGEN_ATTN_FLAT=0 BEVGEN_BENCH_NO_PMC=1
for b in "$@"; do for v in 0 1 0 1; do
  BEVGEN_ATTN_QB_MAJOR=$v python bench.py --steps 3 --warmup 1 --batch $b --no-decode-leg --no-extra-legs --no-cpu-baseline --no-exact-leg 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); l=d['legs']; print('batch $b qb_major=$v', 'scenes/s', round(d['value'],3), 'ms/step', round(d['ms_per_step'],1), 'attention share', round(l['kernel_time_share']['attention'],3), 'TF-equiv', round(l['kernel_tflops']['attention'],1))"
done; done | tee gpurun_out/r05_ab_attn_qb_major.txt
