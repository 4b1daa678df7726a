#!/bin/bash
# MLP weight prefetch into L2: same-box A/B over $BEVGEN_MLP_PREFETCH (0 off, 1 MLP-up pulls MLP-down's image, 2 also MLP-down pulls the next layer's MLP-up image) + traces
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R; : > $O/mlp_pf_ab.txt
for i in 1 2; do for pf in ${PFS:-1 2}; do
  BEVGEN_MLP_PREFETCH=$pf python tools/decode_probe.py 16 2100 fused f16 1 f32,f16 2>/dev/null | grep "ms/step" | sed "s/^/PREFETCH=$pf /" | tee -a $O/mlp_pf_ab.txt
  BEVGEN_MLP_PREFETCH=$pf python tools/decode_probe.py 16 2100 fused f32 1 f32 2>/dev/null | grep "ms/step" | sed "s/^/PREFETCH=$pf /" | tee -a $O/mlp_pf_ab.txt
done; done
for pf in ${PFS:-1 2}; do echo "== PREFETCH=$pf" | tee -a $O/mlp_pf_ab.txt; BEVGEN_MLP_PREFETCH=$pf python tools/decode_trace.py 16 1044 f16 1 f16 2>&1 | grep -v amdgpu.ids | tail -16 | tee -a $O/mlp_pf_ab.txt; done
