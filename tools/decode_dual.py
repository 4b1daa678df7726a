#!/usr/bin/env python
"""Experiment: Route A decode of B sequences as TWO half-batch contexts on two streams (the latency-bound phases of one half overlap the HBM-bound
attention of the other).  usage: decode_dual.py [B=16] [steps=2100] [kv=f32] [weights=f32] [parts=2]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bevgen_amd import presets, synthetic
from bevgen_amd.runtime import Context
from bevgen_amd.weights import gpt_state_dict

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2100
kv = sys.argv[3] if len(sys.argv) > 3 else "f32"
wt = sys.argv[4] if len(sys.argv) > 4 else "f32"
parts = int(sys.argv[5]) if len(sys.argv) > 5 else 2
cfg = presets.config4()
sd = gpt_state_dict(cfg, 1234)
bt = {k: v.cuda() for k, v in synthetic.make_batch(cfg, B, seed=0).items()}

def make(n):
    ctx = Context(cfg, route="ar", max_batch=n, kv_cache=kv, decode_weights=wt)
    ctx.load_state_dict(sd); ctx.set_tables(); ctx.finalize()
    return ctx

one = make(B)
one.ar_sample(bt["cond_ids"], bt["intrinsics_inv"], bt["extrinsics_inv"], steps=8)
torch.cuda.synchronize(); t0 = time.time()
ref = one.ar_sample(bt["cond_ids"], bt["intrinsics_inv"], bt["extrinsics_inv"], steps=steps)
torch.cuda.synchronize(); t1 = time.time() - t0
print(f"single context B={B}: {t1 * 1e3 / steps:.3f} ms/step", flush=True)
one.close()
h = B // parts
ctxs = [make(h) for _ in range(parts)]
streams = [torch.cuda.Stream() for _ in range(parts)]
sl = [{k: v[i * h:(i + 1) * h].contiguous() for k, v in bt.items()} for i in range(parts)]
def run(n):
    outs = []
    for c, s, b in zip(ctxs, streams, sl):
        with torch.cuda.stream(s):
            outs.append(c.ar_sample(b["cond_ids"], b["intrinsics_inv"], b["extrinsics_inv"], steps=n))
    return outs
run(8); torch.cuda.synchronize(); t0 = time.time()
outs = run(steps)
torch.cuda.synchronize(); t2 = time.time() - t0
x = torch.cat(outs, 0)
print(f"{parts} contexts of B={h} on {parts} streams: {t2 * 1e3 / steps:.3f} ms/step for all {B} sequences; tokens equal: {bool(torch.equal(x.cpu(), ref.cpu()))}", flush=True)
