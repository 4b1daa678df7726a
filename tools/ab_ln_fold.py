"""A/B of $BEVGEN_LN_FOLD on the Route-M step (generate + VQGAN decode) at a given batch: prints scenes/s and ms per step.  usage: python tools/ab_ln_fold.py BATCH [STEPS]"""
import sys, time, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import bench
from bevgen_amd import synthetic
B = int(sys.argv[1]); steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
cfg, ctx, _ = bench.build_route_m(6, B, 0, "f16x3", "f32")
bt = {k: v.to(ctx.device) for k, v in synthetic.make_batch(cfg, B, seed=1000).items()}
def step(i):
    ids = ctx.maskgit_generate(bt["cond_ids"], bt["intrinsics_inv"], bt["extrinsics_inv"], timesteps=18, noise_seed=2025 + i, check=False)
    return ctx.vq_decode(ids.reshape(B * 6, -1), latent_hw=(cfg.cam_latent_h, cfg.cam_latent_w), uint8=True, check=False)
step(0); torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps): step(i + 1)
torch.cuda.synchronize(); ctx.synchronize()
dt = (time.perf_counter() - t0) / steps
import os
print(f"LN_FOLD={os.environ.get('BEVGEN_LN_FOLD','default')} batch {B}: {dt*1e3:.2f} ms/step, {B/dt:.3f} scenes/s")
