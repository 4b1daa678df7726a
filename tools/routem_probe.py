#!/usr/bin/env python
"""One Route M transformer forward at the bench shape (B scenes x 6 views, f16x3) + a VQGAN decode of 16 images, for rocprofv3 passes.
usage: routem_probe.py [B=16] [precision=f16x3] [weights=f32|f16]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bevgen_amd import presets, synthetic
from bevgen_amd.runtime import Context
from bevgen_amd.weights import maskgit_state_dict, vq_state_dict

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
precision = sys.argv[2] if len(sys.argv) > 2 else "f16x3"
weights = sys.argv[3] if len(sys.argv) > 3 else "f32"
cfg = presets.config2(6)
dd = presets.VQ_DDCONFIG_F16
ctx = Context(cfg, route="maskgit", vq_ddconfig=dd, vq_n_embed=1024, vq_embed_dim=256, max_batch=B, precision=precision, weights=weights)
ctx.load_state_dict(maskgit_state_dict(cfg, 1234))
ctx.load_state_dict(vq_state_dict(dd, 1024, 256, 99), prefix="first_stage_model.")
ctx.set_tables()
ctx.finalize()
bt = {k: v.cuda() for k, v in synthetic.make_batch(cfg, B, seed=0).items()}
ids = torch.randint(0, cfg.vocab_size + 1, (B * cfg.num_cams, cfg.num_cam_tokens), device="cuda")
for _ in range(2):
    ctx.muse_forward(ids, bt["cond_ids"], bt["intrinsics_inv"], bt["extrinsics_inv"])
ctx.vq_decode(ids[:16].clamp(max=cfg.vocab_size - 1), uint8=True)
torch.cuda.synchronize()
