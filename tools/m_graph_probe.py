#!/usr/bin/env python
"""Experiment: one six-view scene (Route M generate, 18 iterations) launched eagerly vs replayed as ONE hipGraph (captured here with torch.cuda.CUDAGraph around the library
call): how much of the single-scene latency is launch boundary?  usage: m_graph_probe.py [batch]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from bevgen_amd import synthetic

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cams = 6
cfg, ctx, _ = bench.build_route_m(cams, batch, 0, "f16x3")
bt = {k: v.to(ctx.device) for k, v in synthetic.make_batch(cfg, batch, seed=1000).items()}
gen = lambda: ctx.maskgit_generate(bt["cond_ids"], bt["intrinsics_inv"], bt["extrinsics_inv"], timesteps=18, noise_seed=7)
ref = gen(); torch.cuda.synchronize()
t0 = time.time()
for _ in range(5): gen()
torch.cuda.synchronize()
te = (time.time() - t0) / 5
print(f"eager: {te * 1e3:.1f} ms per generate (B = {batch})")
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    gen(); torch.cuda.current_stream().synchronize()
    g = torch.cuda.CUDAGraph()
    t0 = time.time()
    with torch.cuda.graph(g, stream=s):
        out = gen()
    print(f"capture + instantiate: {(time.time() - t0) * 1e3:.0f} ms")
    g.replay(); torch.cuda.current_stream().synchronize()
    t0 = time.time()
    for _ in range(5): g.replay()
    torch.cuda.current_stream().synchronize()
    tg = (time.time() - t0) / 5
print(f"graph replay: {tg * 1e3:.1f} ms per generate; tokens equal to the eager call: {bool(torch.equal(out, ref))}")
