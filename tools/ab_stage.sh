#!/bin/bash
# K/V staging in LDS (ArAttnFusedArgs::stage_cap): parity subset, same-box A/B over $BEVGEN_KV_STAGE, phase traces.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "ar_attn or ln_gemm" 2>&1 | tail -15 > $O/stage_tests.txt
cat $O/stage_tests.txt
if grep -q failed $O/stage_tests.txt; then exit 1; fi
timeout 1500 python -m pytest tests/test_models_gpu.py -x -q -m gpu -k "route_a and not 300_steps and not config5" 2>&1 | tail -4 | tee -a $O/stage_tests.txt
: > $O/stage_ab.txt
for i in 1 2; do
for cap in 0 8 4; do
  BEVGEN_KV_STAGE=$cap python tools/decode_probe.py 16 2100 fused f16 1 f32,f16 2>/dev/null | grep "ms/step" | sed "s/^/STAGE=$cap /" | tee -a $O/stage_ab.txt
done; done
for cap in 0 8; do
  BEVGEN_KV_STAGE=$cap python tools/decode_probe.py 16 2100 fused f32 1 f32 2>/dev/null | grep "ms/step" | sed "s/^/STAGE=$cap /" | tee -a $O/stage_ab.txt
done
: > $O/stage_trace.txt
for cap in 0 8; do for w in f32 f16; do
  echo "== STAGE=$cap" >> $O/stage_trace.txt
  BEVGEN_KV_STAGE=$cap python tools/decode_trace.py 16 1044 f16 1 $w 2>&1 | grep -v amdgpu.ids | head -8 >> $O/stage_trace.txt
done; done
cat $O/stage_trace.txt
