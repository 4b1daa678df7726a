#!/usr/bin/env python
"""What would a decode layer cost if its K/V rows (and weights) came out of the 256 MB memory-side cache instead of HBM?  Route A config 4, B = 16, fp16 cache +
fp16 weights, fused path, with the depth cut to 2 / 3 / 4 / 8 layers: at depth 2 the live K/V rows of ALL layers (2 x 4096 n B x 16 sequences: 34 MB at n = 257,
310 MB at n = 2356) stay resident between steps up to n ~ 1900, at depth 24 nothing does.  Prints per-step time by step bucket and the marginal cost per layer.
usage: mall_probe.py [steps] [depths=2,3,4,8,24]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bevgen_amd import presets, synthetic
from bevgen_amd.runtime import Context
from bevgen_amd.weights import gpt_state_dict

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2100
depths = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "2,3,4,8,24").split(",")]
B = 16
series = {}
for depth in depths:
    cfg = presets.route_a(6, num_layers=depth)
    ctx = Context(cfg, route="ar", max_batch=B, kv_cache="f16", decode_weights="f16", decode_path="fused")
    ctx.load_state_dict(gpt_state_dict(cfg, 1234))
    ctx.set_tables()
    ctx.finalize()
    bt = {k: v.cuda() for k, v in synthetic.make_batch(cfg, B, seed=0).items()}
    ctx.ar_sample(bt["cond_ids"], bt["intrinsics_inv"], bt["extrinsics_inv"], steps=8)
    best = None
    for rep in range(2):
        ctx.ar_step_timing(True)
        ctx.ar_sample(bt["cond_ids"], bt["intrinsics_inv"], bt["extrinsics_inv"], steps=steps)
        torch.cuda.synchronize()
        st = torch.tensor(ctx.ar_step_times(steps + 8))
        ctx.ar_step_timing(False)
        best = st if best is None else torch.minimum(best, st)
    series[depth] = best
    ctx.close()
    print(f"depth {depth}: mean {float(best.mean()):.4f} ms/step", flush=True)
W = 300
print("bucket (steps)   " + "  ".join(f"d{d:>2d} ms/step" for d in depths) + "   | marginal us/layer between consecutive depths")
for i in range(0, steps, W):
    row = [float(series[d][i:i + W].median()) for d in depths]
    marg = [(row[j + 1] - row[j]) * 1e3 / (depths[j + 1] - depths[j]) for j in range(len(depths) - 1)]
    n_mid = 257 + i + W // 2
    print(f"{i:5d}.. n~{n_mid:5d}  " + "  ".join(f"{x:10.4f}" for x in row) + "   | " + "  ".join(f"{x:7.2f}" for x in marg) +
          f"   (K/V per layer {16 * 4096 * n_mid / 1e6:.0f} MB)")
