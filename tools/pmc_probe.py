#!/usr/bin/env python
"""Launch the rooflined kernels with bench-sized operands for `rocprofv3 --pmc` passes (FETCH_SIZE / WRITE_SIZE):
    gemm:    M=24576 (16 scenes x 1536 tokens), N=1024, K=1024, exact-fp32 kernel and the split-precision LDS-DMA kernel
             -> algorithmic bytes = (M*K + N*K + M*N)*4
    decode:  Route A BASELINE config 4 (B=16, H=16, 24 layers, fp32 KV cache), STEPS decode steps through the fused kernels
             -> per launch of ar_attn_fused_kernel, averaged over the steps: 2*B*H*mean_n*64*4 bytes of K/V (+ 786 KB x 16 of q/k/v weights per head)
             -> per launch of skinny_fused_kernel: the weight matrix once ((N*K + M*K + M*N)*4)
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bevgen_amd import presets, synthetic
from bevgen_amd.runtime import Context, _ptr, _stream
from bevgen_amd.weights import gpt_state_dict

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 40
ctx = Context(None)
M, N, K = 24576, 1024, 1024
a = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda")
for _ in range(5):
    ctx.op_gemm(a, w)
out = torch.empty(M, N, device="cuda")
for _ in range(5):
    ctx._check(ctx.lib.bevgen_op_gemm(ctx._h, _ptr(a), _ptr(w), None, None, _ptr(out), M, N, K, 0, 3, _stream()))
torch.cuda.synchronize()
ctx.close()
DENSITY = float(os.environ.get("PMC_DENSITY", "1.0"))   # < 1: per-layer random block layouts (SURVEY 8d config 4 variant); the K/V bytes shrink by the visible fraction
cfg = presets.config4(density=DENSITY) if DENSITY < 1.0 else presets.config4()
ctx = Context(cfg, route="ar", max_batch=16)
sd = gpt_state_dict(presets.config4(), 1234)
visible = 1.0
if DENSITY < 1.0:
    lay_sd, visible = synthetic.random_layer_layouts(cfg)
    sd.update(lay_sd)
    # (the probe decodes the first STEPS rows only: their visible fraction, not the whole decode's)
    import torch as _t
    blk, K = cfg.sparse_block_size, cfg.num_cond_tokens
    tot = vis = 0.0
    for i in range(0, cfg.num_layers, 6):
        lay = lay_sd[f"blocks.{i}.attention.sparse_self_attention.master_layout"]
        for h in range(0, cfg.num_heads, 4):
            for r in range(K, K + STEPS):
                a_row = (cfg.attention_mask[r, : r + 1] != 0)
                tot += float(a_row.sum())
                vis += float((a_row & lay[h, r // blk].bool().repeat_interleave(blk)[: r + 1]).sum())
    visible = vis / tot
ctx.load_state_dict(sd)
ctx.set_tables()
ctx.finalize()
bt = {k: v.cuda() for k, v in synthetic.make_batch(cfg, 16, seed=0).items()}
ctx.ar_sample(bt["cond_ids"], bt["intrinsics_inv"], bt["extrinsics_inv"], steps=STEPS, return_logits=True)   # return_logits: the eager launch path (counter collection + hipGraph replay is unusably slow)
torch.cuda.synchronize()
mean_n = cfg.num_cond_tokens + 1 + (STEPS - 2) / 2.0
open(os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "gpurun_out", "pmc_visible.txt"), "w").write(f"{visible}\n")
print("density", DENSITY, "visible fraction of the causal keys of the probed rows", visible)
print("gemm algorithmic MB", (M * K + N * K + M * N) * 4 / 1e6, "decode steps", STEPS, "mean context", mean_n, "K/V MB per fused launch", 2 * 16 * 16 * mean_n * 64 * 4 / 1e6)
