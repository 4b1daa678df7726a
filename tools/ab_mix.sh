#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R
for i in 1 2; do for top in 2 258 256 264; do
  BEVGEN_KV_STAGE_TOP=$top python tools/decode_probe.py 16 2100 fused f16 1 f32,f16 2>/dev/null | grep "ms/step" | sed "s/^/TOP=$top /" | tee -a $O/mix_ab.txt
done; done
