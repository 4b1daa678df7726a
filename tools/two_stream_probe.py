#!/usr/bin/env python
"""Experiment: the headline step (16 scenes: MaskGit generate + VQGAN decode) as ONE context with 16 scenes vs TWO contexts with 8 scenes each on two streams (scenes are
independent: does the other half's memory-bound work - LayerNorm, softmax phases, GroupNorm - hide under this half's power-bound GEMMs?).  usage: two_stream_probe.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from bevgen_amd import synthetic

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
cams = 6

def make(batch):
    cfg, ctx, _ = bench.build_route_m(cams, batch, 0, "f16x3")
    bt = {k: v.to(ctx.device) for k, v in synthetic.make_batch(cfg, batch, seed=1000).items()}
    return cfg, ctx, bt

def step(cfg, ctx, bt, batch, seed):
    ids = ctx.maskgit_generate(bt["cond_ids"], bt["intrinsics_inv"], bt["extrinsics_inv"], timesteps=18, noise_seed=seed)
    return ctx.vq_decode(ids.reshape(batch * cams, -1), latent_hw=(cfg.cam_latent_h, cfg.cam_latent_w), uint8=True)

cfg, c16, b16 = make(16)
step(cfg, c16, b16, 16, 1); torch.cuda.synchronize()
t0 = time.time()
for i in range(steps): step(cfg, c16, b16, 16, 2 + i)
torch.cuda.synchronize()
t16 = (time.time() - t0) / steps
print(f"one context, 16 scenes: {t16 * 1e3:.1f} ms/step = {16 / t16:.3f} scenes/s")
c16.close()
_, ca, ba = make(8)
_, cb, bb = make(8)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
for c, b, s in ((ca, ba, sa), (cb, bb, sb)):
    with torch.cuda.stream(s): step(cfg, c, b, 8, 1)
torch.cuda.synchronize()
import threading
def worker(c, b, s):
    with torch.cuda.stream(s):
        for i in range(steps): step(cfg, c, b, 8, 2 + i)
t0 = time.time()
ths = [threading.Thread(target=worker, args=a) for a in ((ca, ba, sa), (cb, bb, sb))]   # (ctypes releases the GIL inside the library calls: the two enqueue loops run side by side)
for t in ths: t.start()
for t in ths: t.join()
torch.cuda.synchronize()
t8 = (time.time() - t0) / steps
print(f"two contexts x 8 scenes on two streams, two host threads: {t8 * 1e3:.1f} ms/step = {16 / t8:.3f} scenes/s")
t0 = time.time()
for i in range(steps):
    with torch.cuda.stream(sa): step(cfg, ca, ba, 8, 2 + i)
torch.cuda.synchronize()
t1 = (time.time() - t0) / steps
print(f"one context x 8 scenes alone: {t1 * 1e3:.1f} ms/step = {8 / t1:.3f} scenes/s")
