#!/bin/bash
# A/B of $BEVGEN_GEMM_RME (row-major store form of the plain epilogue of the LDS-DMA GEMM): operator tests, probe, phase trace, then the Route-M step.
BEVGEN_GEMM_RME=1 python -m pytest tests/test_ops_gpu.py -q -k "gemm" 2>&1 | tail -2
for v in 0 1; do
  echo "== RME=$v probe"; BEVGEN_GEMM_RME=$v python tools/gemm_probe.py 3 10 24576,1024,32 24576,1024,1024 24576,3072,1024 2>&1 | grep mode=
  echo "== RME=$v probe (residual)"; PROBE_RESIDUAL=1 BEVGEN_GEMM_RME=$v python tools/gemm_probe.py 3 10 24576,1024,1024 24576,1024,2752 2>&1 | grep mode=
done
for v in 0 1; do echo "== RME=$v trace"; BEVGEN_GEMM_RME=$v BEVGEN_LIB_PATH=$PWD/bevgen_amd/csrc/libbevgen_hip_trace.so python tools/gemm_trace.py 24576,1024,1024 res 2>&1 | grep -v amdgpu.ids; done
for rep in 1 2; do for v in 0 1; do echo -n "RME=$v "; BEVGEN_GEMM_RME=$v python tools/ab_ln_fold.py 16 3 2>&1 | tail -1; done; done
for v in 0 1; do echo -n "RME=$v "; BEVGEN_GEMM_RME=$v python tools/ab_ln_fold.py 4 5 2>&1 | tail -1; done
