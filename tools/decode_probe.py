#!/usr/bin/env python
"""Route A decode probe for rocprofv3 kernel traces: config4 (L=2368, 24 layers), B sequences, `steps` greedy steps (hipGraph path)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bevgen_amd import presets, synthetic
from bevgen_amd.runtime import Context
from bevgen_amd.weights import gpt_state_dict

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
cfg = presets.config4()
ctx = Context(cfg, route="ar", max_batch=B)
ctx.load_state_dict(gpt_state_dict(cfg, 1234))
ctx.set_tables()
ctx.finalize()
bt = {k: v.cuda() for k, v in synthetic.make_batch(cfg, B, seed=0).items()}
ctx.ar_sample(bt["cond_ids"], bt["intrinsics_inv"], bt["extrinsics_inv"], steps=8)
torch.cuda.synchronize()
t0 = time.time()
ctx.ar_sample(bt["cond_ids"], bt["intrinsics_inv"], bt["extrinsics_inv"], steps=steps)
torch.cuda.synchronize()
print(f"B={B} steps={steps} wall {time.time()-t0:.3f}s -> {(time.time()-t0)*1e3/steps:.3f} ms/step")
