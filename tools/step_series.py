#!/usr/bin/env python
"""Per-step times of one graph-replayed config-4 decode (fp16 cache + fp16 weights): bucket means, slowest steps.  usage: step_series.py [B] [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bevgen_amd import presets, synthetic
from bevgen_amd.runtime import Context
from bevgen_amd.weights import gpt_state_dict

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2100
cfg = presets.config4()
ctx = Context(cfg, route="ar", max_batch=B, kv_cache="f16", decode_weights="f16", decode_path="fused")
ctx.load_state_dict(gpt_state_dict(cfg, 1234))
ctx.set_tables()
ctx.finalize()
bt = {k: v.cuda() for k, v in synthetic.make_batch(cfg, B, seed=0).items()}
ctx.ar_sample(bt["cond_ids"], bt["intrinsics_inv"], bt["extrinsics_inv"], steps=8)
for rep in range(2):
    ctx.ar_step_timing(True)
    ctx.ar_sample(bt["cond_ids"], bt["intrinsics_inv"], bt["extrinsics_inv"], steps=steps)
    torch.cuda.synchronize()
    st = torch.tensor(ctx.ar_step_times(steps + 8))
    ctx.ar_step_timing(False)
    print(f"run {rep}: {st.numel()} steps, mean {float(st.mean()):.4f} median {float(st.median()):.4f} ms")
    for i in range(0, st.numel(), 150):
        b = st[i:i + 150]
        print(f"  steps {i:4d}..: mean {float(b.mean()):.4f} median {float(b.median()):.4f} min {float(b.min()):.4f} max {float(b.max()):.4f}")
    top = torch.topk(st, 12)
    print("  slowest:", [(int(i), round(float(v), 3)) for v, i in zip(top.values, top.indices)])
ctx.close()
