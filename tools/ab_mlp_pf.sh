#!/bin/bash
# MLP-up prefetching MLP-down's weight image into L2: same-box A/B over $BEVGEN_MLP_PREFETCH + traces
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R; : > $O/mlp_pf_ab.txt
for i in 1 2; do for pf in 0 1; do
  BEVGEN_MLP_PREFETCH=$pf python tools/decode_probe.py 16 2100 fused f16 1 f32,f16 2>/dev/null | grep "ms/step" | sed "s/^/PREFETCH=$pf /" | tee -a $O/mlp_pf_ab.txt
  BEVGEN_MLP_PREFETCH=$pf python tools/decode_probe.py 16 2100 fused f32 1 f32 2>/dev/null | grep "ms/step" | sed "s/^/PREFETCH=$pf /" | tee -a $O/mlp_pf_ab.txt
done; done
for pf in 0 1; do echo "== PREFETCH=$pf" | tee -a $O/mlp_pf_ab.txt; BEVGEN_MLP_PREFETCH=$pf python tools/decode_trace.py 16 1044 f16 1 f16 2>&1 | grep -v amdgpu.ids | tail -8 | tee -a $O/mlp_pf_ab.txt; done
