#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
: > $O/stage_ab3.txt
for i in 1 2; do
for combo in "8 2" "8 256" "8 257" "8 258" "8 1"; do set -- $combo
  BEVGEN_KV_STAGE=$1 BEVGEN_KV_STAGE_TOP=$2 python tools/decode_probe.py 16 2100 fused f16 1 f32,f16 2>/dev/null | grep "ms/step" | sed "s/^/STAGE=$1 TOP=$2 /" | tee -a $O/stage_ab3.txt
done; done
for combo in "8 256" "8 258"; do set -- $combo
  BEVGEN_KV_STAGE=$1 BEVGEN_KV_STAGE_TOP=$2 python tools/decode_probe.py 16 2100 fused f32 1 f32 2>/dev/null | grep "ms/step" | sed "s/^/STAGE=$1 TOP=$2 /" | tee -a $O/stage_ab3.txt
done
