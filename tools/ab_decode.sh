#!/bin/bash
# Same-box A/B of the Route A decode step: this tree's library vs another build (.ab/lib<name>.so; r02 = the round-2 library built from commit 9bfa2ec), alternating,
# full decodes.   usage on the GPU box: bash tools/ab_decode.sh [steps=2100] [other=r02]
R=${GRAFT_REPO_ROOT:-/root/repo}
STEPS=${1:-2100}; OTHER=${2:-r02}
for i in 1 2; do
for lib in new $OTHER; do
  if [ $lib = new ]; then unset BEVGEN_LIB_PATH; else export BEVGEN_LIB_PATH=$R/.ab/lib$lib.so; fi
  for mode in "f32 f32" "f16 f32" "f16 f16"; do set -- $mode
    python $R/tools/decode_probe.py 16 $STEPS fused $1 1 $2 2>/dev/null | grep "ms/step" | sed "s/^/$lib /"
  done
done; done
