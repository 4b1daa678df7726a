#!/bin/bash
# same-box A/B of the stride-1 convolution variant (MODE_CONV3S): $BEVGEN_CONV_FAST=0 (general kernel) / 1, optionally against another build in .ab/lib<name>.so
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/r05_ab_conv_fast.txt; : > $OUT
for i in 1 2 3; do
  for v in 0 1; do BEVGEN_CONV_FAST=$v python tools/vq_probe.py 96 f16x3 5 2>/dev/null | sed "s/^/CONV_FAST=$v /" | tee -a $OUT; done
  for l in "$@"; do BEVGEN_LIB_PATH=$PWD/.ab/lib$l.so python tools/vq_probe.py 96 f16x3 5 2>/dev/null | sed "s/^/lib=$l /" | tee -a $OUT; done
done
for v in 0 1; do BEVGEN_CONV_FAST=$v python tools/vq_probe.py 6 f16x3 5 2>/dev/null | sed "s/^/CONV_FAST=$v /" | tee -a $OUT; done
BEVGEN_CONV_FAST=1 timeout 600 python -m pytest tests -m gpu -x -q -k "vq or conv" 2>&1 | tail -3 | tee -a $OUT
