#!/usr/bin/env python
"""One-off randomized parity sweep of the LDS-DMA 3x3 convolution through bevgen_op_conv3x3 (kernel='dma' = the stride-1 variant MODE_CONV3S, 'dma_general' = the
general variant) against fp64 and against each other (bit for bit): ragged image sizes, channel counts that are any multiple of 32, 1-700 output channels, with / without
residual.  usage on the GPU box: python tools/fuzz/conv_fuzz.py [cases=60] [seed=0]"""
import math, os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from bevgen_amd.runtime import Context

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ctx = Context(None)
worst = (0.0, None)
for case in range(n_cases):
    n = rng.choice([1, 1, 2, 3, 5])
    H, W = rng.choice([1, 2, 3, 7, 14, 16, 25, 32, 57, 64]), rng.choice([1, 2, 5, 16, 25, 31, 32, 64, 100])
    if rng.random() < 0.15:
        n, H, W = rng.choice([2, 4]), rng.choice([96, 128]), rng.choice([128, 200])   # enough rows for 256-row tiles
    Cin = 32 * rng.choice([1, 2, 3, 4, 5, 8])
    Cout = rng.choice([1, 3, 32, 64, 100, 128, 129, 256, 384, 700])
    if n * H * W * Cout > 40e6:
        Cout = 64
    g = torch.Generator().manual_seed(5000 + case)
    x = torch.randn(n, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)
    b = torch.randn(Cout, generator=g)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    res = torch.randn_like(ref, dtype=torch.float32) if rng.random() < 0.5 else None
    if res is not None:
        ref = ref + res.double()
    args = (x.permute(0, 2, 3, 1).contiguous().cuda(), w.permute(0, 2, 3, 1).contiguous().cuda(), b.cuda())
    r = None if res is None else res.permute(0, 2, 3, 1).contiguous().cuda()
    fast = ctx.op_conv3x3(*args, residual=r, kernel="dma")
    gen = ctx.op_conv3x3(*args, residual=r, kernel="dma_general")
    err = float((fast.cpu().permute(0, 3, 1, 2).double() - ref).abs().max() / ref.abs().max())
    same = bool(torch.equal(fast, gen))
    tag = f"case {case}: n={n} H={H} W={W} Cin={Cin} Cout={Cout} residual={res is not None}"
    if err > worst[0]:
        worst = (err, tag)
    if err > 2e-6 or not same:
        print("FAIL", tag, "rel err", err, "fast == general:", same)
        sys.exit(1)
print(f"{n_cases} cases ok; worst rel err {worst[0]:.2e} ({worst[1]})")
ctx.close()
