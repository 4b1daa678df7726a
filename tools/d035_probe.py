#!/usr/bin/env python
"""Route A decode at density 0.35 (per-layer random block layouts, SURVEY 8d config-4 variant): ms per step through the product path.
usage: d035_probe.py [steps=1000] [kv=f16] [weights=f32]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bevgen_amd import presets, synthetic
from bevgen_amd.runtime import Context
from bevgen_amd.weights import gpt_state_dict

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
kv = sys.argv[2] if len(sys.argv) > 2 else "f16"
wt = sys.argv[3] if len(sys.argv) > 3 else "f32"
cfg = presets.config4(density=0.35)
sd = dict(gpt_state_dict(presets.config4(), 1234))
lay_sd, vis = synthetic.random_layer_layouts(cfg)
sd.update(lay_sd)
ctx = Context(cfg, route="ar", max_batch=16, kv_cache=kv, decode_weights=wt, decode_path="fused")
ctx.load_state_dict(sd); ctx.set_tables(); ctx.finalize()
bt = {k: v.cuda() for k, v in synthetic.make_batch(cfg, 16, seed=0).items()}
ctx.ar_sample(bt["cond_ids"], bt["intrinsics_inv"], bt["extrinsics_inv"], steps=8)
torch.cuda.synchronize(); t0 = time.time()
x = ctx.ar_sample(bt["cond_ids"], bt["intrinsics_inv"], bt["extrinsics_inv"], steps=steps)
torch.cuda.synchronize(); dt = time.time() - t0
print(f"density 0.35 kv={kv} weights={wt} steps={steps}: {dt * 1e3 / steps:.3f} ms/step (incl. prefill); token checksum {int(x.sum())}")
ctx.close()
