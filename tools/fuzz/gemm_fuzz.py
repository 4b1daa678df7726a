#!/usr/bin/env python
"""One-off randomized parity sweep of the LDS-DMA split-precision GEMM through bevgen_op_gemm (modes 3 / 4 / 5) against fp64: ragged M, N, few k-tiles.
usage on the GPU box: python tools/fuzz/gemm_fuzz.py [cases=80] [seed=0]   (honours BEVGEN_GEMM_STAGES)"""
import math, os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bevgen_amd.runtime import Context, _ptr, _stream

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 80
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ctx = Context(None)
worst = (0.0, None)
for case in range(n_cases):
    M = rng.choice([1, 7, 31, 64, 127, 128, 129, 255, 300, 513, 1000, 1537, 3000]) + rng.randint(0, 5)
    N = rng.choice([1, 4, 60, 64, 128, 130, 200, 256, 700, 1024]) + rng.randint(0, 3)
    K = 32 * rng.choice([1, 2, 3, 4, 5, 7, 8, 13, 32])
    if os.environ.get("FUZZ_BIG"):   # problems large enough for 256-row tiles and the row-split launches (tile counts just above a multiple of 256)
        N = rng.choice([1024, 1000, 2048, 5504, 640])
        gx = (N + 127) // 128
        gy = (256 * rng.choice([1, 2, 3]) + gx - 1) // gx + rng.choice([0, 1, 2, 5])
        M = 256 * gy - rng.randint(0, 255)
        K = 32 * rng.choice([1, 2, 4, 9])
    mode = rng.choice([3, 3, 4, 5])
    if mode == 5 and (K // 32 < 6 or os.environ.get("FUZZ_BIG")):   # (split-K is the small-M path: the library refuses it for 256-row problems)
        mode = 3
    g = torch.Generator().manual_seed(case)
    a = torch.randn(M, K, generator=g) * 3.0
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    if mode == 4:
        w = w.half().float()
    use_epi = rng.random() < 0.5
    b = torch.randn(N, generator=g) if use_epi else None
    r = torch.randn(M, N, generator=g) if use_epi else None
    ref = a.double() @ w.double().t()
    if use_epi:
        ref = torch.nn.functional.gelu(ref + b.double()) + r.double()
    da, dw = a.cuda(), w.cuda()
    db = b.cuda() if use_epi else None
    dr = r.cuda() if use_epi else None
    out = torch.full((M, N), float("nan"), device="cuda")
    ctx._check(ctx.lib.bevgen_op_gemm(ctx._h, _ptr(da), _ptr(dw), _ptr(db), _ptr(dr), _ptr(out), M, N, K, 1 if use_epi else 0, mode, _stream()))
    o = out.cpu().double()
    assert torch.isfinite(o).all(), (M, N, K, mode, "non-finite output")
    err = float((o - ref).norm() / ref.norm().clamp_min(1e-30))
    if err > worst[0]:
        worst = (err, (M, N, K, mode, use_epi))
    assert err < 3e-6, (M, N, K, mode, use_epi, err)
print("cases", n_cases, "worst", worst)
