cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "key_ranges" 2>&1 | tail -5
for b in 1 2 3 4 6; do for v in 0 -1; do
  if [ $v = 0 ]; then export BEVGEN_ATTN_FLAT=0; else unset BEVGEN_ATTN_FLAT; fi
  BEVGEN_BENCH_NO_PMC=1 python bench.py --steps 3 --warmup 1 --batch $b --no-decode-leg --no-extra-legs --no-cpu-baseline --no-exact-leg 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); l=d['legs']; print('batch $b flat=$v', 'scenes/s', round(d['value'],3), 'ms/step', round(d['ms_per_step'],1), {k: round(x,3) for k,x in l['kernel_time_share'].items()}, {k: round(x,1) for k,x in l['kernel_tflops'].items()})"
done; done | tee gpurun_out/r05_ab_attn_flat.txt
