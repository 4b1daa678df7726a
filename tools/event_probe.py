import os, sys
sys.path.insert(0, "/root/repo")
import torch
from bevgen_amd import presets, synthetic
from bevgen_amd.runtime import Context
from bevgen_amd.weights import gpt_state_dict
cfg = presets.config4()
ctx = Context(cfg, route="ar", max_batch=16, kv_cache="f16", decode_weights="f16", decode_path="fused")
ctx.load_state_dict(gpt_state_dict(cfg, 1234)); ctx.set_tables(); ctx.finalize()
bt = {k: v.cuda() for k, v in synthetic.make_batch(cfg, 16, seed=0).items()}
ctx.ar_sample(bt["cond_ids"], bt["intrinsics_inv"], bt["extrinsics_inv"], steps=8)
ctx.profile_begin()
ctx.ar_sample(bt["cond_ids"], bt["intrinsics_inv"], bt["extrinsics_inv"], steps=2100)
torch.cuda.synchronize()
p = ctx.profile_end()
for k in ("decode_attention", "gemm_skinny"):
    print(k, p[k], "avg us", p[k]["ms"] * 1e3 / max(p[k]["launches"], 1))
