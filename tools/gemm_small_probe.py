#!/usr/bin/env python
"""Small-M split-precision GEMM launches for a rocprofv3 --kernel-trace run: M = rows of one scene (1536), a list of (N, K), REPS launches each, in order.
tools/gemm_small_report.py reads the trace database and prints the duration per shape (the LDS-DMA GEMM launches in start order, REPS per shape)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bevgen_amd.runtime import Context

REPS = 20
M = int(os.environ.get("PROBE_M", "1536"))
SHAPES = [(1024, 256), (1024, 512), (1024, 1024), (1024, 2048), (1024, 4096), (2048, 1024), (512, 1024), (4096, 1024)]
if os.environ.get("PROBE_SHAPES"):   # "NxK,NxK,..."
    SHAPES = [tuple(int(v) for v in t.split("x")) for t in os.environ["PROBE_SHAPES"].split(",")]
ctx = Context(None)
for N, K in SHAPES:
    a = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda")
    for _ in range(REPS):
        ctx.op_gemm(a, w, skinny=3)
    torch.cuda.synchronize()
print("M", M, "reps", REPS, "shapes", SHAPES)
