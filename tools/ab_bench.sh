for i in 1 2; do
for lib in new old; do
  if [ $lib = old ]; then export BEVGEN_LIB_PATH=$GRAFT_REPO_ROOT/.ab/libold.so; else unset BEVGEN_LIB_PATH; fi
  python bench.py --steps 3 --warmup 1 --no-decode-leg --no-extra-legs --no-cpu-baseline --no-exact-leg 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib', round(d['value'],3), round(d['ms_per_step'],1), 'it', round(d['ms_per_maskgit_iteration'],2), 'vq/scene', round(d['vqgan_decode_ms_per_scene'],2), {k: round(v,3) for k,v in d['kernel_time_share'].items()})"
done; done
