#!/bin/bash
# A/B of $BEVGEN_GEMM_RPF (residual prefetch in the k loop's tail of the LDS-DMA GEMM): operator probe with a residual, then the Route-M step at 16 / 1 / 2 scenes.
for v in 0 1; do
  echo "== RPF=$v probe (residual)"; PROBE_RESIDUAL=1 BEVGEN_GEMM_RPF=$v python tools/gemm_probe.py 3 10 24576,1024,32 24576,1024,1024 24576,1024,2752 2>&1 | grep mode=
done
for rep in 1 2; do for v in 0 1; do echo -n "RPF=$v "; BEVGEN_GEMM_RPF=$v python tools/ab_ln_fold.py 16 3 2>&1 | tail -1; done; done
for v in 0 1; do echo -n "RPF=$v "; BEVGEN_GEMM_RPF=$v python tools/ab_ln_fold.py 1 10 2>&1 | tail -1; done
for v in 0 1; do echo -n "RPF=$v "; BEVGEN_GEMM_RPF=$v python tools/ab_ln_fold.py 2 6 2>&1 | tail -1; done
BEVGEN_GEMM_RPF=1 python -m pytest tests/test_ops_gpu.py -q -k "gemm" 2>&1 | tail -2
