#!/bin/bash
# Same-box A/B on the GPU box, alternating the variants so that box-to-box spread (+-2.5 %) and clock drift cancel.  One script for what were 15 one-offs:
#
#   bash tools/ab.sh <workload> <axis> <spec> [extra args]
#     workload  m       Route M headline step (bench.py, 3 steps, no side legs)             extra args -> bench.py
#               decode  Route A config-4 decode, B = 16 (tools/decode_probe.py)             extra: STEPS (default 2100), MODES (default "f16:f16 f16:f32 f32:f32" = kv:weights)
#     axis      lib     this tree's library vs other builds:  spec = "name [name ...]"  ->  .ab/lib<name>.so   (build the other tree, copy its .so there; .ab/ travels
#                       with the gpurun snapshot and is git-ignored)
#               env     an environment switch:                 spec = "VAR=v1,v2[,v3 ...]"
#   ROUNDS (default 2 for decode, 3 for m) repetitions; results also appended to gpurun_out/ab_<workload>.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
W=$1; AXIS=$2; SPEC=$3; shift 3
case $AXIS in
  lib) VARIANTS="new $SPEC" ;;
  env) VAR=${SPEC%%=*}; VARIANTS=$(echo ${SPEC#*=} | tr ',' ' ') ;;
  *) echo "axis must be lib or env"; exit 2 ;;
esac
select_variant() {
  if [ $AXIS = lib ]; then
    if [ $1 = new ]; then unset BEVGEN_LIB_PATH; else export BEVGEN_LIB_PATH=$R/.ab/lib$1.so; [ -f $BEVGEN_LIB_PATH ] || { echo "missing $BEVGEN_LIB_PATH"; exit 2; }; fi
    TAG=$1
  else export $VAR=$1; TAG="$VAR=$1"; fi
}
OUT=$O/ab_$W.txt; : > $OUT
if [ $W = m ]; then
  export BEVGEN_BENCH_NO_PMC=1
  for i in $(seq ${ROUNDS:-3}); do for v in $VARIANTS; do
    select_variant $v
    python bench.py --steps 3 --warmup 1 --no-decode-leg --no-extra-legs --no-cpu-baseline --no-exact-leg "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); l=d['legs']; print('$TAG', 'scenes/s', round(d['value'],3), 'ms/step', round(d['ms_per_step'],1), 'it', round(l['ms_per_maskgit_iteration'],2), 'vq/scene', round(l.get('vqgan_decode_ms_per_scene',0),2), {k: round(x,3) for k,x in l['kernel_time_share'].items()}, {k: round(x,1) for k,x in l['kernel_tflops'].items()})" | tee -a $OUT
  done; done
elif [ $W = decode ]; then
  STEPS=${1:-2100}; MODES=${2:-"f16:f16 f16:f32 f32:f32"}
  for i in $(seq ${ROUNDS:-2}); do for v in $VARIANTS; do
    select_variant $v
    for mode in $MODES; do
      python tools/decode_probe.py 16 $STEPS fused ${mode%%:*} 1 ${mode##*:} 2>/dev/null | grep "ms/step" | sed "s/^/$TAG /" | tee -a $OUT
    done
  done; done
else echo "workload must be m or decode"; exit 2; fi
