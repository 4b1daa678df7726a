#!/bin/bash
# Same-box A/B of the Route A decode step: this tree's library vs .ab/libr02.so (the round-2 library built from commit 9bfa2ec), alternating, full 2100-step decodes.
# usage on the GPU box: bash tools/ab_decode.sh [steps=2100]
R=${GRAFT_REPO_ROOT:-/root/repo}
STEPS=${1:-2100}
for i in 1 2; do
for lib in new r02; do
  if [ $lib = r02 ]; then export BEVGEN_LIB_PATH=$R/.ab/libr02.so; else unset BEVGEN_LIB_PATH; fi
  for mode in "f32 f32" "f16 f32" "f16 f16"; do set -- $mode
    python $R/tools/decode_probe.py 16 $STEPS fused $1 1 $2 2>/dev/null | grep "ms/step" | sed "s/^/$lib /"
  done
done; done
