#!/usr/bin/env python
"""GEMM probe for PMC / timing: bench-sized Route M projections through bevgen_op_gemm.  usage: gemm_probe.py [mode: 0 fp32 | 2 split | 3 split LDS-DMA | 4 LDS-DMA with f16 weights (two MFMAs per product)] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bevgen_amd.runtime import Context, _ptr, _stream

mode = int(sys.argv[1]) if len(sys.argv) > 1 else 2
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ctx = Context(None)
shapes = [(24576, 1024, 1024), (24576, 5460, 1024), (24576, 1024, 2752)]
if len(sys.argv) > 3:   # extra shapes "M,N,K M,N,K ..."
    shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[3:]]
for (M, N, K) in shapes:
    a = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.03
    if mode == 4: w = w.half().float()
    out = torch.empty(M, N, device="cuda")
    res = _ptr(out) if os.environ.get("PROBE_RESIDUAL") else None   # x = x + proj, the shape of the residual-stream projections
    def run():
        ctx._check(ctx.lib.bevgen_op_gemm(ctx._h, _ptr(a), _ptr(w), None, res, _ptr(out), M, N, K, 0, mode, _stream()))
    run(); torch.cuda.synchronize()
    ctx.profile_begin()
    for _ in range(reps): run()
    torch.cuda.synchronize()
    pe = ctx.profile_end()
    dt = (pe["gemm"]["ms"] + pe["gemm_small"]["ms"]) * 1e-3 / reps   # (small-problem, stream-K and persistent launches are recorded as gemm_small)   # (a GEMM may be two launches: the LDS-DMA kernel's tail split)
    if res is not None:
        out.zero_(); run(); torch.cuda.synchronize()
    ref = (a[:64].double() @ w.double().t()).float()
    err = ((out[:64] - ref).abs().max() / ref.abs().max()).item()
    print(f"mode={mode} M={M} N={N} K={K}: {dt*1e6:.0f} us  {2*M*N*K/dt/1e12:.1f} TF (GEMM kernel only, HIP events)  rel err of 64 rows vs fp64 {err:.2e}")
