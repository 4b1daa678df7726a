#!/usr/bin/env python
"""Phase timeline of the fused Route A decode kernels (device timestamps per workgroup, last launch of each kind).
usage: decode_trace.py [B] [steps] [kv] [samples_per_layout] [weights]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bevgen_amd import presets, synthetic
from bevgen_amd.runtime import Context
from bevgen_amd.weights import gpt_state_dict

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1044
kv = sys.argv[3] if len(sys.argv) > 3 else "f32"
S = int(sys.argv[4]) if len(sys.argv) > 4 else 1
WT = sys.argv[5] if len(sys.argv) > 5 else "f32"
cfg = presets.config4()
ctx = Context(cfg, route="ar", max_batch=B, kv_cache=kv, decode_weights=WT, decode_path="fused")
ctx.load_state_dict(gpt_state_dict(cfg, 1234))
ctx.set_tables()
ctx.finalize()
bt = {k: v.repeat_interleave(S, dim=0).cuda() for k, v in synthetic.make_batch(cfg, B // S, seed=0).items()}
ctx.ar_sample(bt["cond_ids"], bt["intrinsics_inv"], bt["extrinsics_inv"], steps=8, samples_per_layout=S)
ctx.trace_begin()
ctx.ar_sample(bt["cond_ids"], bt["intrinsics_inv"], bt["extrinsics_inv"], steps=steps, samples_per_layout=S)
tr = ctx.trace_end().double() / 100.0   # us (100 MHz)
names = [("ln1+qkv+attention", ["start", "ln1 done", "qkv done", "append done", "attention done", "end"]),
         # (the fused MLP launch - ar_mlp_fused_kernel, the default at D = 1024 - stamps six points into this slot; the two-launch form the first four)
         ("ln2+MLP-up" if os.environ.get("BEVGEN_LN2_FOLD") == "0" else "fused MLP launch (ln2 + up + GELU | exchange | down)",
          ["start", "A tile staged", "MFMA+reduce", "end"] if os.environ.get("BEVGEN_LN2_FOLD") == "0" else
          ["start", "rows staged", "up-projection + GELU stored", "exchange passed", "down-projection partial sums in LDS", "end"]),
         ("MLP-down", ["start", "A tile staged", "MFMA+reduce", "end"])]
print(f"B={B} S={S} kv={kv} weights={WT}: last step context n={cfg.num_cond_tokens + steps - 1}")
for k, (name, pts) in enumerate(names):
    t = tr[k]
    used = t[:, 0] > 0
    t = t[used]
    if t.numel() == 0:
        continue
    t0 = t[:, 0].min()
    print(f"{name}: {t.shape[0]} workgroups; first start -> last end {float(t[:, len(pts) - 1].max() - t0):.2f} us; start skew {float(t[:, 0].max() - t0):.2f} us")
    if k == 0:
        print(f"    x rows arrived + first barrier at mean {float((t[:, 6] - t0).mean()):.2f} us; statistics done at mean {float((t[:, 7] - t0).mean()):.2f} us")
    for i in range(1, len(pts)):
        d = t[:, i] - t[:, i - 1]
        print(f"    {pts[i - 1]:>16} -> {pts[i]:<16} mean {float(d.mean()):6.2f}  min {float(d.min()):6.2f}  max {float(d.max()):6.2f} us   (done at mean {float((t[:, i] - t0).mean()):6.2f})")
ctx.close()
