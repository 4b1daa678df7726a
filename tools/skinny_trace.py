#!/usr/bin/env python
"""Phase timeline of one fused skinny-GEMM launch (device timestamps), cold vs warm weights.  usage: skinny_trace.py M N K ln"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bevgen_amd.runtime import Context

M, N, K, ln = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
ctx = Context(None)
a = torch.randn(M, K, device="cuda")
g, b = torch.randn(K, device="cuda"), torch.randn(K, device="cuda")
ws = [torch.randn(N, K, device="cuda") * 0.03 for _ in range(24)]
pts = ["start", "A tile staged", "MFMA+reduce", "end"]
for mode in ("cold", "warm", "warm"):
    seq = ws if mode == "cold" else [ws[0]] * 4
    for w in seq:
        ctx.op_ln_gemm(a, w, ln_w=g if ln else None, ln_b=b if ln else None)
    torch.cuda.synchronize()
    ctx.trace_begin()
    ctx.op_ln_gemm(a, seq[0], ln_w=g if ln else None, ln_b=b if ln else None)
    t = ctx.trace_end().double()[1] / 100.0
    t = t[t[:, 0] > 0]
    t0 = t[:, 0].min()
    print(f"{mode}: {t.shape[0]} workgroups, first start -> last end {float(t[:, 3].max() - t0):.2f} us, start skew {float(t[:, 0].max() - t0):.2f}; " +
          "; ".join(f"{pts[i]} at {float((t[:, i] - t0).mean()):.2f}" for i in range(1, 4)) + f"; loads issued at {float((t[:, 4] - t0).mean()):.2f}; A arrived at {float((t[:, 5] - t0).mean()):.2f}; wave 7 started at {float((t[:, 7] - t0).mean()):.2f}, its A arrived at {float((t[:, 6] - t0).mean()):.2f}", flush=True)
