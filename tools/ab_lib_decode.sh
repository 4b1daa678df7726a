#!/bin/bash
# Same-box A/B of the Route A decode step: this tree's library vs another build (.ab/lib<name>.so), alternating; op tests of the changed kernels first; traces last.
# usage on the GPU box: bash tools/ab_lib_decode.sh [other=head] [test -k expression]
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R
OTHER=${1:-head}; KEXPR=${2:-"ar_attn or ln_gemm"}
timeout 1200 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "$KEXPR" 2>&1 | tail -3
: > $O/lib_decode_ab.txt
for i in 1 2; do for lib in new $OTHER; do
  if [ $lib = new ]; then unset BEVGEN_LIB_PATH; else export BEVGEN_LIB_PATH=$R/.ab/lib$lib.so; fi
  python tools/decode_probe.py 16 2100 fused f16 1 f32,f16 2>/dev/null | grep "ms/step" | sed "s/^/$lib /" | tee -a $O/lib_decode_ab.txt
  python tools/decode_probe.py 16 2100 fused f32 1 f32 2>/dev/null | grep "ms/step" | sed "s/^/$lib /" | tee -a $O/lib_decode_ab.txt
done; done
unset BEVGEN_LIB_PATH
python tools/decode_trace.py 16 1044 f16 1 f16 2>&1 | grep -v amdgpu.ids | tail -8 | tee -a $O/lib_decode_ab.txt
