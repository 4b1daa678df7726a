#!/usr/bin/env python
"""Route A decode step: chains A/B.  BASELINE config 4, B sequences, `steps` greedy steps through the hipGraph path, decode_chains in {1, 2, 4}.
usage: chain_probe.py [B] [steps] [chains=1,2,4] [modes=f32:f32,f16:f32,f16:f16] [samples_per_layout=1]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bevgen_amd import presets, synthetic
from bevgen_amd.runtime import Context
from bevgen_amd.weights import gpt_state_dict

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
chains = [int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "1,2,4").split(",")]
modes = [m.split(":") for m in (sys.argv[4] if len(sys.argv) > 4 else "f32:f32,f16:f32,f16:f16").split(",")]
S = int(sys.argv[5]) if len(sys.argv) > 5 else 1
paths = (sys.argv[6] if len(sys.argv) > 6 else "fused").split(",")
cfg = presets.config4()
sd = gpt_state_dict(cfg, 1234)
bt = {k: v.repeat_interleave(S, dim=0).cuda() for k, v in synthetic.make_batch(cfg, B // S, seed=0).items()}
for kv, wt in modes:
    ref = None
    for nch, path in [(n_, p_) for p_ in paths for n_ in chains]:
        ctx = Context(cfg, route="ar", max_batch=B, kv_cache=kv, decode_weights=wt, decode_chains=nch, decode_path=path)
        ctx.load_state_dict(sd)
        ctx.set_tables()
        ctx.finalize()
        ctx.ar_sample(bt["cond_ids"], bt["intrinsics_inv"], bt["extrinsics_inv"], steps=8, samples_per_layout=S)
        torch.cuda.synchronize()
        ctx.ar_step_timing(True)
        t0 = time.time()
        x = ctx.ar_sample(bt["cond_ids"], bt["intrinsics_inv"], bt["extrinsics_inv"], steps=steps, samples_per_layout=S)
        torch.cuda.synchronize()
        dt = time.time() - t0
        st = ctx.ar_step_times(steps + 8)
        x = x.cpu()
        same = "" if ref is None else f"  tokens == chains {chains[0]}: {bool(torch.equal(x, ref))}"
        if ref is None:
            ref = x
        import numpy as np
        print(f"B={B} S={S} steps={steps} kv={kv} w={wt} path={path} chains={nch}: {dt * 1e3 / steps:.3f} ms/step incl. prefill; per replay median {np.median(st):.3f} p99 {np.percentile(st, 99):.3f} "
              f"first100 {st[:100].mean():.3f} last100 {st[-100:].mean():.3f}{same}", flush=True)
        ctx.close()
