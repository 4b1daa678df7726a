#!/bin/bash
# ln1 folded into the projection of the fused decode kernel: parity, same-box A/B against the library before (.ab/libhead.so), traces
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "ar_attn or ln_gemm" 2>&1 | tail -15 > $O/fold_tests.txt
cat $O/fold_tests.txt
if grep -q failed $O/fold_tests.txt; then exit 1; fi
timeout 1500 python -m pytest tests/test_models_gpu.py tests/test_dropin_gpu.py -x -q -m gpu -k "route_a or gpt or ar_ or config4 or config5 or config1" 2>&1 | tail -6 | tee -a $O/fold_tests.txt
: > $O/fold_ab.txt
for i in 1 2; do
for lib in new head; do
  if [ $lib = new ]; then unset BEVGEN_LIB_PATH; else export BEVGEN_LIB_PATH=$R/.ab/lib$lib.so; fi
  python tools/decode_probe.py 16 2100 fused f16 1 f32,f16 2>/dev/null | grep "ms/step" | sed "s/^/$lib /" | tee -a $O/fold_ab.txt
  python tools/decode_probe.py 16 2100 fused f32 1 f32 2>/dev/null | grep "ms/step" | sed "s/^/$lib /" | tee -a $O/fold_ab.txt
done; done
unset BEVGEN_LIB_PATH
python tools/decode_probe.py 64 600 fused f32 4 f32 2>/dev/null | grep "ms/step" | sed "s/^/new config5-like /" | tee -a $O/fold_ab.txt
BEVGEN_LIB_PATH=$R/.ab/libhead.so python tools/decode_probe.py 64 600 fused f32 4 f32 2>/dev/null | grep "ms/step" | sed "s/^/head config5-like /" | tee -a $O/fold_ab.txt
for w in f32 f16; do python tools/decode_trace.py 16 1044 f16 1 $w 2>&1 | grep -v amdgpu.ids | head -8 | tee -a $O/fold_trace.txt; done
