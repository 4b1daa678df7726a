#!/bin/bash
# K/V staging A/B: cap x top combos, fp16 KV with both weight types, then traces of the best
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
: > $O/stage_ab2.txt
for i in 1 2; do
for combo in "0 0" "4 4" "6 4" "8 4" "8 6" "8 8" "8 2"; do set -- $combo
  BEVGEN_KV_STAGE=$1 BEVGEN_KV_STAGE_TOP=$2 python tools/decode_probe.py 16 2100 fused f16 1 f32,f16 2>/dev/null | grep "ms/step" | sed "s/^/STAGE=$1 TOP=$2 /" | tee -a $O/stage_ab2.txt
done; done
BEVGEN_KV_STAGE=8 BEVGEN_KV_STAGE_TOP=4 python tools/decode_probe.py 16 2100 fused f32 1 f32 2>/dev/null | grep "ms/step" | sed "s/^/STAGE=8 TOP=4 /" | tee -a $O/stage_ab2.txt
BEVGEN_KV_STAGE=0 python tools/decode_probe.py 16 2100 fused f32 1 f32 2>/dev/null | grep "ms/step" | sed "s/^/STAGE=0 /" | tee -a $O/stage_ab2.txt
: > $O/stage_trace2.txt
for combo in "8 4" "8 8" "4 4"; do set -- $combo
  echo "== STAGE=$1 TOP=$2" >> $O/stage_trace2.txt
  BEVGEN_KV_STAGE=$1 BEVGEN_KV_STAGE_TOP=$2 python tools/decode_trace.py 16 1044 f16 2>&1 | grep -v amdgpu.ids | head -8 >> $O/stage_trace2.txt
done
cat $O/stage_trace2.txt
