#!/usr/bin/env python
"""Decode-step projection shapes through the skinny GEMM (HIP-event time per launch): usage skinny_probe.py [M]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bevgen_amd.runtime import Context

M = int(sys.argv[1]) if len(sys.argv) > 1 else 16
ctx = Context(None)
for (N, K) in [(3072, 1024), (4096, 1024), (1024, 4096), (1024, 1024)]:
    a = torch.randn(M, K, device="cuda")
    ws = [torch.randn(N, K, device="cuda") * 0.03 for _ in range(24)]   # 24 different weight matrices: every launch streams from HBM
    for w in ws[:2]:
        ctx.op_gemm(a, w, skinny=True)
    torch.cuda.synchronize()
    ctx.profile_begin()
    for w in ws:
        ctx.op_gemm(a, w, skinny=True)
    torch.cuda.synchronize()
    p = ctx.profile_end()["gemm_skinny"]
    us = p["ms"] * 1e3 / p["launches"]
    print(f"M={M} N={N} K={K}: {us:.2f} us per launch, {N * K * 4 / us / 1e6:.2f} TB/s of weights")
