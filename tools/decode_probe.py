#!/usr/bin/env python
"""Route A decode probe: BASELINE config 4 (L=2368, 24 layers), B sequences, `steps` greedy steps through the hipGraph path.
usage: decode_probe.py [B] [steps] [paths=fused,per_op] [kv=f32,f16] [samples_per_layout=1] [weights=f32] [precision=fp32 (prefill arithmetic: fp32 | f16x3)]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bevgen_amd import presets, synthetic
from bevgen_amd.runtime import Context
from bevgen_amd.weights import gpt_state_dict

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
paths = (sys.argv[3] if len(sys.argv) > 3 else "fused,per_op").split(",")
kvs = (sys.argv[4] if len(sys.argv) > 4 else "f32,f16").split(",")
S = int(sys.argv[5]) if len(sys.argv) > 5 else 1
weights = (sys.argv[6] if len(sys.argv) > 6 else "f32").split(",")
precision = sys.argv[7] if len(sys.argv) > 7 else "fp32"
cfg = presets.config4()
sd = gpt_state_dict(cfg, 1234)
layouts = B // S
bt = {k: v.repeat_interleave(S, dim=0).cuda() for k, v in synthetic.make_batch(cfg, layouts, seed=0).items()}
ref = None
for kv, wt in [(k, w) for k in kvs for w in weights]:
    for path in paths:
        ctx = Context(cfg, route="ar", max_batch=B, kv_cache=kv, decode_path=path, decode_weights=wt, precision=precision)
        ctx.load_state_dict(sd)
        ctx.set_tables()
        ctx.finalize()
        ctx.ar_sample(bt["cond_ids"], bt["intrinsics_inv"], bt["extrinsics_inv"], steps=8, samples_per_layout=S)
        torch.cuda.synchronize()
        t0 = time.time()
        x = ctx.ar_sample(bt["cond_ids"], bt["intrinsics_inv"], bt["extrinsics_inv"], steps=steps, samples_per_layout=S)
        torch.cuda.synchronize()
        dt = time.time() - t0
        x = x.cpu()
        same = "" if ref is None else f" tokens equal to first run: {bool(torch.equal(x, ref))} ({(x != ref).sum().item()} differ)"
        if ref is None:
            ref = x
        print(f"B={B} S={S} steps={steps} kv={kv} weights={wt} path={path} precision={precision}: wall {dt:.3f}s -> {dt * 1e3 / steps:.3f} ms/step (incl. prefill){same}", flush=True)
        ctx.close()
