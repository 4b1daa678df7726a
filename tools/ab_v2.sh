#!/bin/bash
# second form of the fused decode kernel: parity subset, same-box A/B ($BEVGEN_DECODE_V2, $BEVGEN_KV_STAGE), phase traces
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "ar_attn or ln_gemm" 2>&1 | tail -15 > $O/v2_tests.txt
cat $O/v2_tests.txt
if grep -q failed $O/v2_tests.txt; then exit 1; fi
timeout 1500 python -m pytest tests/test_models_gpu.py -x -q -m gpu -k "route_a and not 300_steps and not config5" 2>&1 | tail -8 | tee -a $O/v2_tests.txt
: > $O/v2_ab.txt
for i in 1 2; do
for combo in "0 0" "1 0" "1 1"; do set -- $combo
  BEVGEN_DECODE_V2=$1 BEVGEN_KV_STAGE=$2 python tools/decode_probe.py 16 2100 fused f16 1 f32,f16 2>/dev/null | grep "ms/step" | sed "s/^/V2=$1 STAGE=$2 /" | tee -a $O/v2_ab.txt
done; done
for combo in "0 0" "1 1"; do set -- $combo
  BEVGEN_DECODE_V2=$1 BEVGEN_KV_STAGE=$2 python tools/decode_probe.py 16 2100 fused f32 1 f32 2>/dev/null | grep "ms/step" | sed "s/^/V2=$1 STAGE=$2 /" | tee -a $O/v2_ab.txt
done
: > $O/v2_trace.txt
for combo in "1 1" "1 0"; do set -- $combo
  echo "== V2=$1 STAGE=$2" >> $O/v2_trace.txt
  BEVGEN_DECODE_V2=$1 BEVGEN_KV_STAGE=$2 python tools/decode_trace.py 16 1044 f16 2>&1 | grep -v amdgpu.ids | head -8 >> $O/v2_trace.txt
done
cat $O/v2_trace.txt
