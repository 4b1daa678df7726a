#!/bin/bash
# Same-box A/B of the Route A decode step over an environment switch:  bash tools/ab_env.sh VAR "v1 v2 ..." [steps=2100] [modes="f32:f32 f16:f32 f16:f16"]
R=${GRAFT_REPO_ROOT:-/root/repo}
VAR=$1; VALS=$2; STEPS=${3:-2100}; MODES=${4:-"f32:f32 f16:f32 f16:f16"}
for i in 1 2; do
for v in $VALS; do
  export $VAR=$v
  for mode in $MODES; do
    python $R/tools/decode_probe.py 16 $STEPS fused ${mode%%:*} 1 ${mode##*:} 2>/dev/null | grep "ms/step" | sed "s/^/$VAR=$v /"
  done
done; done
