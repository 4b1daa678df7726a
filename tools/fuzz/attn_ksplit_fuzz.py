#!/usr/bin/env python
"""One-off randomized parity sweep of the key-split form of the split-precision attention kernel (bevgen_op_attention_ex) against fp64 and against the unsplit kernel.
usage on the GPU box: python tools/fuzz/attn_ksplit_fuzz.py [cases=40] [seed=0]"""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bevgen_amd.runtime import Context

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ctx = Context(None, precision="f16x3")
worst = (0.0, None)
for case in range(n_cases):
    B, H = rng.choice([1, 2, 3]), rng.choice([1, 2, 5])
    Nq = rng.choice([1, 31, 32, 33, 200, 256, 257, 600])
    Nk = rng.choice([32, 33, 64, 100, 257, 512, 1000, 1568])
    Nk_pad = (Nk + 31) // 32 * 32
    splits = rng.randint(2, min(8, Nk_pad // 32)) if Nk_pad // 32 >= 2 else 1
    g = torch.Generator().manual_seed(1000 + case)
    q = torch.randn(B, H, Nq, 64, generator=g); k = torch.randn(B, H, Nk, 64, generator=g); v = torch.randn(B, H, Nk, 64, generator=g)
    kp = torch.zeros(B, H, Nk_pad, 64); kp[:, :, :Nk] = k
    vp = torch.zeros(B, H, Nk_pad, 64); vp[:, :, :Nk] = v
    scale = 0.2
    bias_real = torch.randn(Nq, Nk, generator=g) * 2
    bias_real[torch.rand(Nq, Nk, generator=g) < 0.3] = -1e30
    bias_real[:, rng.randrange(Nk)] = 0.25   # every row sees at least one key
    sim = torch.einsum("bhid,bhjd->bhij", q.double(), k.double()) * scale + bias_real.double()
    ref = torch.einsum("bhij,bhjd->bhid", sim.softmax(-1), v.double()).permute(0, 2, 1, 3).reshape(B, Nq, H * 64)
    bias = torch.full((Nq, Nk_pad), -1e30); bias[:, :Nk] = bias_real
    dq, dk, dv, db = q.cuda(), kp.cuda(), vp.cuda(), bias.cuda()
    out = ctx.op_attention(dq, dk, dv, db, scale, key_splits=splits).cpu().double()
    assert torch.isfinite(out).all(), (B, H, Nq, Nk, splits)
    err = float((out - ref).norm() / ref.norm())
    if err > worst[0]:
        worst = (err, (B, H, Nq, Nk, splits))
    assert err < 5e-6, (B, H, Nq, Nk, splits, err)
print("cases", n_cases, "worst", worst)
