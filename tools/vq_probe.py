#!/usr/bin/env python
"""VQGAN decode alone at the bench shape: n images (default 96 = 16 six-view scenes) of 16 x 16 latents -> 256 x 256 uint8; ms per scene over a few repetitions.
usage: vq_probe.py [n=96] [precision=f16x3] [reps=5]     """
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bevgen_amd import presets
from bevgen_amd.runtime import Context
from bevgen_amd.weights import vq_state_dict

n = int(sys.argv[1]) if len(sys.argv) > 1 else 96
precision = sys.argv[2] if len(sys.argv) > 2 else "f16x3"
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dd = presets.VQ_DDCONFIG_F16
ctx = Context(None, vq_ddconfig=dd, vq_n_embed=1024, vq_embed_dim=256, precision=precision)
ctx.load_state_dict(vq_state_dict(dd, 1024, 256, 99), prefix="first_stage_model.")
ctx.finalize()
ids = torch.randint(0, 1024, (n, 256), device="cuda", generator=torch.Generator(device="cuda").manual_seed(0))
ctx.vq_decode(ids, uint8=True)
torch.cuda.synchronize()
ts = []
for _ in range(reps):
    t0 = time.time()
    out = ctx.vq_decode(ids, uint8=True)
    torch.cuda.synchronize()
    ts.append(time.time() - t0)
import hashlib
pix = ctx.vq_decode(ids[:12], uint8=False)   # fp32 pixels of two scenes: a bit-level fingerprint for A/B runs over kernel variants that must not change results
sha = hashlib.sha256(pix.cpu().numpy().tobytes()).hexdigest()[:16]
print(f"n={n} precision={precision}: {min(ts) * 1e3 / (n / 6):.3f} ms per six-view scene (best of {reps}; mean {sum(ts) / len(ts) * 1e3 / (n / 6):.3f}), checksum {int(out.long().sum())}, fp32 pixels sha256 {sha}")
ctx.close()
