#!/usr/bin/env python
"""Decode-step projection through the fused skinny kernel, for rocprofv3 kernel traces: `cold` = 24 different weight matrices (every launch
streams from HBM), `warm` = one matrix re-used (L2 / Infinity-Cache resident).  usage: skinny_probe.py M cold|warm N K ln"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bevgen_amd.runtime import Context

M, mode, N, K, ln = int(sys.argv[1]), sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
ctx = Context(None)
a = torch.randn(M, K, device="cuda")
g, b = torch.randn(K, device="cuda"), torch.randn(K, device="cuda")
ws = [torch.randn(N, K, device="cuda") * 0.03 for _ in range(24)]
seq = ws if mode == "cold" else [ws[0]] * 24
for rep in range(10):
    for w in seq:
        ctx.op_ln_gemm(a, w, ln_w=g if ln else None, ln_b=b if ln else None)
torch.cuda.synchronize()
