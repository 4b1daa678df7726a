#!/bin/bash
# Same-box A/B of the Route A decode step over an environment switch, all three storage modes: bash tools/ab_env_decode.sh VAR "v1 v2" [steps=2100]
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R
VAR=$1; VALS=$2; STEPS=${3:-2100}
: > $O/env_decode_ab.txt
for i in 1 2; do for v in $VALS; do
  export $VAR=$v
  python tools/decode_probe.py 16 $STEPS fused f16 1 f32,f16 2>/dev/null | grep "ms/step" | sed "s/^/$VAR=$v /" | tee -a $O/env_decode_ab.txt
  python tools/decode_probe.py 16 $STEPS fused f32 1 f32 2>/dev/null | grep "ms/step" | sed "s/^/$VAR=$v /" | tee -a $O/env_decode_ab.txt
done; done
