#!/usr/bin/env python
"""bench.py - BEVGen stage-2 sampling path on MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1: the script launches its own N ranks (one per GPU, `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`)
unless it already runs under a launcher (WORLD_SIZE set); inside, the RCCL world size must equal N or the run fails.

Workload (BASELINE.json configs[1]): Route M (muse_stage_two bidirectional MaskGit decoder, released hyper-parameters: 14 layers, D=1024, 16 heads,
18 iterations, top-k thres 0.9, gumbel / critic noise drawn in the sampler kernels, self token critic), 6 views of 256x256, batch = 16 scenes per GPU: BEV token grid -> MaskGit generate -> VQGAN decode
-> uint8 pixels.  A "step" is one batch of 16 scenes.  Inputs (BEV token ids, camera matrices) and random-init weights of the reference
architecture are synthetic and resident in HBM before the timed region.  N > 1: independent scenes are sharded over ranks (weak scaling, 16 scenes
per GPU), no data-path collective; the only RCCL traffic is the final gather of the uint8 pixels to rank 0 (inside the timed region); a
`strong_scaling` leg (16 scenes in total) is added to the line.

Default arithmetic mode: f16x3 (every GEMM / conv / attention product as three f16 MFMAs on hi/lo splits, fp32 accumulation: fp32-class accuracy,
bit-identical greedy tokens on every parity fixture); the line also carries a one-step `exact_fp32_mode` leg (exact fp32 MFMA).

One SHORT JSON line on rank 0: the contract keys, `roofline` for the dominant kernel of the workload (the LDS-DMA split-precision GEMM; `traffic` measured by
this run through two `rocprofv3 --pmc` child passes, tools/pmc_inrun.py), `cpu_baseline` (the CPU oracle timed on this host), and LAST a flat `legs` object with the
scalar result of every other leg - at N = 1: Route A decode (BASELINE config 4: ms per decode step mean / median / p99, decode-attention roofline fraction by
SURVEY 8(d)'s bytes and by storage bytes, whole-step fraction; f32 / f16 KV cache, f16 weights, density 0.35, split path), config 5 on one GPU, the released 3-camera
shape, single-scene latency, exact-fp32 and f16-weights modes; at N > 1: the strong-scaling leg and BASELINE configs[4] across the ranks (64 sequences per GPU,
token ids gathered to rank 0), per-rank step times.  The unabridged per-leg objects go to gpurun_out/bench_detail_n<N>.json (`detail_file`).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8.0 TB/s spec
MFMA_FP32_PEAK_TF = 157.3   # v_mfma_f32_32x32x2_f32: exact fp32 at the vector rate (dense)
MFMA_F16_PEAK_TF = 2500.0   # dense f16/bf16 MFMA peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=16, help="scenes per GPU per step")
    ap.add_argument("--cams", type=int, default=6)
    ap.add_argument("--timesteps", type=int, default=18)
    ap.add_argument("--cpu-baseline", default="sample", choices=["sample", "full", "none"], help="sample = one whole scene with 4 of the 18 MaskGit iterations (about 30 s of CPU work), full = all 18 (minutes)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-decode-leg", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the config 5 / 3-camera legs")
    ap.add_argument("--decode-batch", type=int, default=16)
    ap.add_argument("--decode-steps", type=int, default=0, help="0 = a full decode (N image tokens)")
    ap.add_argument("--precision", default="f16x3", choices=["fp32", "f16x3"], help="fp32 = exact fp32 MFMA; f16x3 = split-precision products (both bit-exact on the parity fixtures)")
    ap.add_argument("--no-exact-leg", action="store_true", help="skip the additional one-step run in exact-fp32 mode")
    return ap.parse_args()


# Control-flow dry run (tests/test_bench_cpu.py): BEVGEN_BENCH_DRYRUN=1 swaps the library context for a shape-only stub and RCCL for gloo so that the N > 1 paths of this
# file (rank sharding, barrier / max-over-ranks timing, the gather, the strong-scaling leg, rank-0 reporting) can be executed on a CPU box before the first real multi-GPU
# run.  The line it prints carries "dry_run": true and no performance meaning; nothing in the product path reads this variable.
DRY_RUN = os.environ.get("BEVGEN_BENCH_DRYRUN") == "1"


class _CpuEvent:
    def __init__(self, enable_timing=True):
        self.t = None

    def record(self):
        self.t = time.perf_counter()

    def elapsed_time(self, other):
        return (other.t - self.t) * 1e3


class _StubContext:
    """Shape-only stand-in for bevgen_amd.runtime.Context (dry run): token ids / pixels of the right shape and dtype, no arithmetic."""

    def __init__(self, cfg):
        import torch
        self.cfg, self.device = cfg, torch.device("cpu")

    def maskgit_generate(self, cond_ids, I_inv, E_inv, timesteps=18, noise_seed=0, **kw):
        import torch
        B = cond_ids.shape[0]
        g = torch.Generator().manual_seed(int(noise_seed) % (2 ** 31))
        return torch.randint(0, self.cfg.vocab_size, (B * self.cfg.num_cams, self.cfg.cam_latent_h, self.cfg.cam_latent_w), generator=g)

    def vq_decode(self, ids, latent_hw=None, uint8=False, **kw):
        import torch
        return (ids.reshape(ids.shape[0], -1)[:, :1, None, None] % 256).to(torch.uint8).expand(ids.shape[0], 3, 16 * latent_hw[0], 16 * latent_hw[1]).contiguous()

    def ar_sample(self, cond_ids, I_inv, E_inv, steps=None, **kw):
        import torch
        g = torch.Generator().manual_seed(int(cond_ids.sum()) % (2 ** 31))
        return torch.randint(0, self.cfg.vocab_size, (cond_ids.shape[0], self.cfg.num_cams, self.cfg.num_cam_tokens), generator=g)

    def profile_begin(self):
        pass

    def profile_end(self):
        return {k: {"launches": 0.0, "ms": 0.0, "work": 0.0} for k in ("gemm", "conv3x3", "attention", "decode_attention", "gemm_skinny", "gemm_small")}

    def close(self):
        pass


def self_launch(args):
    """--gpus N without a launcher: become `torch.distributed.run` with N ranks on this node."""
    import torch

    have = args.gpus if DRY_RUN else torch.cuda.device_count()
    if have < args.gpus:
        print(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) are visible", file=sys.stderr)
        sys.exit(2)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    os.execv(sys.executable, cmd)


def gemm_algorithmic_bytes(cfg, batch):
    """Mean algorithmic bytes per launch of the seven projections of a Route M layer (A + W + C, 4 bytes per element: operands are hi/lo f16 plane pairs)."""
    M, D = batch * cfg.num_img_tokens, cfg.num_embed
    F = int(D * 4 * 2 / 3)
    Fp = (F + 31) // 32 * 32
    shapes = [(D, D)] * 4 + [(2 * D, D), (2 * Fp, D), (D, Fp)]   # to_q / to_out (self, cross), to_kv, GEGLU up (C leaves half as wide), down
    tot = 0.0
    for n, k in shapes:
        c = Fp if n == 2 * Fp else n
        tot += 4.0 * (M * k + n * k + M * c)
    return tot / len(shapes)


def inrun_traffic(kernel_substr):
    """`roofline.traffic`, measured by this run: tools/pmc_inrun.py spawns two `rocprofv3 --kernel-trace --pmc` child passes (FETCH_SIZE, WRITE_SIZE) over one
    transformer forward of the headline workload - outside the timed region, rank 0, N = 1.  $BEVGEN_BENCH_NO_PMC=1 skips it (traffic = null)."""
    if DRY_RUN or os.environ.get("BEVGEN_BENCH_NO_PMC") == "1":
        return {"error": "skipped"}
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import pmc_inrun
        return pmc_inrun.measure(kernel_substr)
    except Exception as e:   # a failed counter pass must not take the bench line with it
        return {"error": f"{type(e).__name__}: {e}"}


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/*_pmc_hbm_traffic.json: FETCH_SIZE x2 + WRITE_SIZE, gfx950
    correction applied) at the probe shape recorded there (detail file only: the headline `roofline.traffic` is measured in-run, see inrun_traffic)."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_hbm_traffic.json")), reverse=True):
        try:
            d = json.load(open(path))["kernels"]
        except Exception:
            continue
        for k, v in d.items():
            if k.split("<")[0] in kernel:
                return {"bytes_per_launch": v["traffic_bytes"], "algorithmic_bytes": v["algorithmic_bytes"], "shape": v["shape"], "source": os.path.basename(path)}
    return None


def pct(xs, q):
    import numpy as np
    return float(np.percentile(np.asarray(xs, dtype=np.float64), q)) if len(xs) else None


_SD_CACHE = {}


def cached(key, make):
    """Synthetic weights are regenerated from (seed, parameter name) on the host: seconds per model, so every leg of one run shares them."""
    if key not in _SD_CACHE:
        _SD_CACHE[key] = make()
    return _SD_CACHE[key]


def build_route_m(cams, batch, device, precision="fp32", weights="f32"):
    from bevgen_amd import presets

    cfg = presets.config2(cams)
    if DRY_RUN:
        return cfg, _StubContext(cfg), None
    from bevgen_amd.runtime import Context
    from bevgen_amd.weights import maskgit_state_dict, vq_state_dict

    sd = cached(("maskgit", cams), lambda: maskgit_state_dict(cfg, 1234))
    dd = presets.VQ_DDCONFIG_F16
    ctx = Context(cfg, route="maskgit", vq_ddconfig=dd, vq_n_embed=1024, vq_embed_dim=256, device=device, max_batch=batch, precision=precision, weights=weights)
    ctx.load_state_dict(sd)
    ctx.load_state_dict(cached(("vq",), lambda: vq_state_dict(dd, 1024, 256, 99)), prefix="first_stage_model.")
    ctx.set_tables()
    ctx.finalize()
    return cfg, ctx, sd


# ----------------------------------------------------------------------------------------------------------------- CPU baseline
def cpu_info():
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return model, os.cpu_count() or 1


def tune_threads(run):
    """The torch thread count that is fastest on this host for `run` (a bounded piece of the workload itself: on many-core hosts "all cores" is far from
    the optimum at these sizes, and a GEMM micro-benchmark does not predict the attention / normalisation parts)."""
    import torch

    best_t, best = 1, float("inf")
    for nt in [n for n in (8, 16, 32, 64, 128) if n <= (os.cpu_count() or 1)] or [1]:
        torch.set_num_threads(nt)
        t0 = time.time()
        run()
        dt = time.time() - t0
        if dt < best:
            best_t, best = nt, dt
    torch.set_num_threads(best_t)
    return best_t


def cpu_baseline(cams, timesteps, mode):
    """Oracle (CPU restatement of the reference algorithm, `kind: port`) on this host, on a bounded sample of the headline workload (about 30 s of CPU
    work): ONE whole scene, measured end to end - MaskGit generate + VQGAN decode of the `cams` images - with 4 MaskGit iterations instead of 18
    (7 transformer forwards instead of 35; the iterations are identical in cost), scaled to 18 iterations.  mode 'full' runs all 18 (minutes).
    Beside it: the reference's own schedule (4 forwards per iteration, derived from the measured per-forward time), a 1-thread figure on a bounded
    piece, and the Route A legs (BASELINE config 4, one sequence: prefill + 8 KV-cache steps vs one full-recompute step of the reference loop)."""
    import torch
    from bevgen_amd import presets, synthetic
    from oracle import cases, restate as R

    model, cores = cpu_info()
    cfg = presets.config2(cams)
    sd = cases.maskgit_state_dict(cfg, 1234)
    bt = synthetic.make_batch(cfg, 1, seed=0)
    dd = presets.VQ_DDCONFIG_F16
    sdv = cases.vq_state_dict(dd, 1024, 256, 99)
    idm = torch.full((cams, cfg.num_cam_tokens), cfg.vocab_size, dtype=torch.long)

    def two_layers():
        with torch.no_grad():
            R.muse_forward(sd, cfg, idm, bt["cond_ids"], bt["intrinsics_inv"], bt["extrinsics_inv"], depth=2, heads=cfg.num_heads)

    two_layers()
    threads = tune_threads(two_layers)
    its = timesteps if mode == "full" else min(4, timesteps)
    with torch.no_grad():
        t0 = time.time()
        ids = R.maskgit_generate(sd, cfg, bt["cond_ids"], bt["intrinsics_inv"], bt["extrinsics_inv"], depth=cfg.num_layers, heads=cfg.num_heads, timesteps=its)
        t_gen = time.time() - t0
        t0 = time.time()
        R.vq_decode_ids(sdv, dd, ids.reshape(cams, -1), (cfg.cam_latent_h, cfg.cam_latent_w))
        t_dec = time.time() - t0
    n_meas, n_fwd = 2 * its - 1, 2 * timesteps - 1
    t_fwd = t_gen / n_meas
    scene = t_fwd * n_fwd + t_dec
    out = {"value": 1.0 / scene, "unit": "scenes/s", "cores": threads, "kind": "port",
           "kind_note": ("the repo's CPU restatement of the reference algorithm (oracle/restate.py, PyTorch CPU fp32), same schedule as the GPU path (35 forwards); "
                         + ("all iterations measured" if its == timesteps else f"BOUNDED SAMPLE: {its} of the {timesteps} MaskGit iterations measured and scaled linearly")
                         + "; `reference_schedule_value` (72 forwards) is DERIVED from the measured per-forward time, not run"),
           "sample": (f"1 whole scene ({cams}x256x256) measured end to end: MaskGit generate with {its} iterations = {n_meas} transformer forwards ({t_gen:.1f} s) + VQGAN decode of {cams} images "
                      f"({t_dec:.1f} s)" + ("" if its == timesteps else f"; scaled to {timesteps} iterations = {n_fwd} forwards of the same cost")),
           "host": {"cpu_model": model, "logical_cores": cores, "threads_used": threads, "threads_note": "fastest of 8..128 torch threads on two transformer layers of this workload"},
           "seconds_per_forward": t_fwd, "seconds_vqgan_decode_per_image": t_dec / cams,
           "reference_schedule_value": 1.0 / (4 * timesteps * t_fwd + t_dec),
           "reference_schedule_note": f"the reference runs {4 * timesteps} forwards per scene (CFG null + critic double forwards, muse_net:272-276, 394-396); derived from the measured per-forward time"}
    # 1-thread figure on a bounded piece: the first 2 of the 14 layers of one six-view forward
    torch.set_num_threads(1)
    t0 = time.time()
    two_layers()
    t1 = time.time() - t0
    out["one_thread"] = {"seconds_2_of_14_layers": t1, "scenes_per_s_extrapolated": 1.0 / (t1 * (cfg.num_layers / 2.0) * n_fwd),
                         "note": "1 torch thread, 2 transformer layers of one six-view forward, scaled by 7 x 35 (transformer only)"}
    torch.set_num_threads(threads)
    # Route A: BASELINE config 4, one sequence
    del sd
    cfg4 = presets.config4()
    sd4 = cases.gpt_state_dict(cfg4, 1234)
    b4 = synthetic.make_batch(cfg4, 1, seed=0)
    with torch.no_grad():
        t0 = time.time()
        cache = R.ARCache(sd4, cfg4, b4["cond_ids"], b4["intrinsics_inv"], b4["extrinsics_inv"])
        t_pre = time.time() - t0
        del cache
        t0 = time.time()
        R.ar_sample_cached(sd4, cfg4, b4["cond_ids"], b4["intrinsics_inv"], b4["extrinsics_inv"], steps=8)
        t_c8 = time.time() - t0 - t_pre
        t0 = time.time()
        R.ar_sample_full_recompute(sd4, cfg4, b4["cond_ids"], b4["intrinsics_inv"], b4["extrinsics_inv"], steps=1)
        t_full = time.time() - t0
    out["route_a_config4"] = {"prefill_s": t_pre, "kv_cache_ms_per_step": max(t_c8, 0.0) * 1e3 / 8, "full_recompute_ms_per_step": t_full * 1e3,
                              "sample": "1 sequence, L=2368: prefill + 8 KV-cache steps; 1 step of the reference's full L-token forward per token (ar_lm:172-219)"}
    return out


# ----------------------------------------------------------------------------------------------------------------- Route A legs
def route_a_bytes(cfg, B, steps, kv_bytes, G=1, w_bytes_per=4):
    """Algorithmic HBM bytes of a decode run: K and V rows of the context once per (sequence, head, layer) - the shared condition prefix once
    per group - and every weight matrix once per step."""
    H, D, K, Lyr, V = cfg.num_heads, cfg.num_embed, cfg.num_cond_tokens, cfg.num_layers, cfg.vocab_size
    kv = 0.0
    for s in range(steps):
        n = K + s + 1
        kv += 2.0 * H * 64 * kv_bytes * Lyr * (B * (n - (K if G > 1 else 0)) + (B // G) * (K if G > 1 else 0))
    w = steps * float(w_bytes_per) * (Lyr * (3 * D * D + 8 * D * D) + V * D)
    return kv, w


def decode_leg(device, batch, steps, kv_cache="f32", S=1, top_k=None, stochastic=False, weights="f32", density=1.0, path="fused"):
    """Route A (BASELINE config 4: nuScenes 6-view 224x400, 24 layers, L=2368, blk 16, camera bias): decode of `steps` tokens for `batch` sequences through
    the product path (hipGraph replay of the fused decode step).  S > 1 = BASELINE config 5: groups of S samples share their BEV layout.
    density < 1 = SURVEY 8(d) config 4 variant: every attention layer gets its own random per-head block layouts (drawn like the reference does at construction,
    gpt:176, maskgen:217-228) - absent blocks are never read.  path = 'split': LayerNorm + QKV projection kernel, then the attention-only kernel."""
    import torch
    from bevgen_amd import presets, synthetic
    from bevgen_amd.runtime import Context
    from bevgen_amd.weights import gpt_state_dict

    cfg = presets.config4(density=density) if density < 1.0 else presets.config4()
    ctx = Context(cfg, route="ar", device=device, max_batch=batch, kv_cache=kv_cache, decode_weights=weights, decode_path=path)
    sd = dict(cached(("gpt", "config4"), lambda: gpt_state_dict(presets.config4(), 1234)))
    visible_frac = 1.0
    if density < 1.0:
        lay_sd, visible_frac = synthetic.random_layer_layouts(cfg)
        sd.update(lay_sd)
    ctx.load_state_dict(sd)
    ctx.set_tables()
    ctx.finalize()
    bt = synthetic.make_batch(cfg, batch // S, seed=0)
    bt = {k: v.repeat_interleave(S, dim=0).to(ctx.device) for k, v in bt.items()}
    steps = steps or cfg.num_img_tokens
    noise = synthetic.uniform_noise((steps, batch), 2025, 1).to(ctx.device) if stochastic else None

    def sample(n_steps):
        kw = dict(samples_per_layout=S)
        if stochastic:
            kw.update(greedy=False, top_k=top_k, temperature=1.0, noise_u=noise[:n_steps].contiguous() if n_steps != steps else noise)
        return ctx.ar_sample(bt["cond_ids"], bt["intrinsics_inv"], bt["extrinsics_inv"], steps=n_steps, **kw)

    sample(8)  # warm-up
    torch.cuda.synchronize()
    t0 = time.time()
    ctx.ar_prefill(bt["cond_ids"], bt["intrinsics_inv"], bt["extrinsics_inv"])   # the K condition rows of every sequence through the 24 layers
    torch.cuda.synchronize()
    prefill_ms = (time.time() - t0) * 1e3
    # pass 1: the product path with one event per replayed step -> wall time, per-step distribution
    ctx.ar_step_timing(True)
    t0 = time.time()
    sample(steps)
    torch.cuda.synchronize()
    wall = time.time() - t0
    st = ctx.ar_step_times(steps + 8)
    ctx.ar_step_timing(False)
    # pass 2: same work launched eagerly with a HIP-event pair ATTACHED to every fused attention / skinny-GEMM launch (hipExtLaunchKernelGGL start / stop events = the
    # kernel's own begin / end, what rocprofv3 --kernel-trace reports) -> per-kernel rooflines
    ctx.profile_begin()
    sample(steps)
    torch.cuda.synchronize()
    prof = ctx.profile_end()
    # pass 3: phase timestamps of the last step's kernels (attention phase alone)
    ctx.trace_begin()
    sample(min(steps, 1044))
    tr = ctx.trace_end().double()[0] / 100.0
    tr = tr[tr[:, 0] > 0]
    n_last = cfg.num_cond_tokens + min(steps, 1044) - 1
    kvb = 4 if kv_cache == "f32" else 2
    phase_us = float((tr[:, 4] - tr[:, 3]).mean()) if tr.numel() else None
    K = cfg.num_cond_tokens
    phase_bytes = visible_frac * 2.0 * cfg.num_heads * 64 * kvb * (batch * (n_last - (K if S > 1 else 0)) + (batch // S) * (K if S > 1 else 0))
    ctx.close()
    da, gs = prof["decode_attention"], prof["gemm_skinny"]
    ach = da["work"] / (da["ms"] * 1e-3) / 1e9 if da["ms"] > 0 else 0.0
    kv_bytes, w_bytes = route_a_bytes(cfg, batch, steps, kvb, S, 2 if weights == "f16" else 4)
    kv_bytes *= visible_frac     # block-sparse layouts: only the rows of present blocks are algorithmic traffic
    ach *= visible_frac
    step_ach = (kv_bytes + w_bytes) / wall / 1e9
    # what the fused launch actually streams: this layer's K/V rows (storage bytes, visible blocks only) PLUS its q / k / v weight rows (3 D^2 elements, once per launch, through
    # the XCD L2s) - the fraction a prologue-bound launch (short or sparse contexts, fp32 K/V) should be judged by next to the SURVEY 8(d) figure, which counts K/V alone
    qkv_w_bytes = 0.0 if path == "split" else 3.0 * cfg.num_embed * cfg.num_embed * (2 if weights == "f16" else 4) * da["launches"]
    ach_kvw = (da["work"] * visible_frac + qkv_w_bytes) / (da["ms"] * 1e-3) / 1e9 if da["ms"] > 0 else 0.0
    out = {
        "ms_per_decode_step": wall * 1e3 / steps, "ms_per_decode_step_median": pct(st, 50), "ms_per_decode_step_p99": pct(st, 99), "decode_prefill_ms": prefill_ms,
        "roofline_decode_attention": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                                      "frac_by_survey_8d_fp16_bytes": ach / HBM_PEAK_GBS * (2.0 / kvb),   # SURVEY 8(d) counts 98 304 n bytes per sequence-step (fp16 K/V) whatever the storage
                                      "frac_kv_plus_qkv_weight_bytes": ach_kvw / HBM_PEAK_GBS,

                                      "traffic": pmc_traffic("ar_attn_fused_kernel") if kv_cache == "f32" and weights == "f32" and S == 1 else None, "kernel": "ar_attn_fused_kernel (ln1 + q/k/v projection + decode attention in one launch; achieved = K/V bytes / WHOLE kernel time)",
                                      # (no bandwidth is derived from this window any more: since the K/V staging, the leading pieces of every wave's key walk are requested
                                      #  by LDS-DMA during the prologue, i.e. BEFORE the window's first timestamp - bytes / window came out above the HBM peak)
                                      "attention_phase": {"us": phase_us, "context": n_last, "kv_bytes_of_the_launch": phase_bytes, "staged_pieces_per_wave": "up to 8 x 1 KiB, requested before the window opens",
                                                          "note": "duration of the key-walk phase alone (device timestamps of the last launch); a duration, not a bandwidth"},
                                      "launches": int(da["launches"]), "avg_us": da["ms"] * 1e3 / max(da["launches"], 1),
                                      "config": f"Route A config4: B={batch}, H=16, all contexts 257..{256 + steps} of the decode, {kv_cache} KV cache, {weights} projection weights, L=2368" + (f", {S} samples per layout (shared prefix read once per group)" if S > 1 else "")},
        "decode_step_roofline": {"bound": "hbm", "achieved": step_ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": step_ach / HBM_PEAK_GBS,
                                 "bytes_per_step": {"kv": kv_bytes / steps, "weights": w_bytes / steps}, "note": f"(K/V rows of the context + every {weights} weight matrix once) per step / wall time per step"},
        "decode_sequences_per_s": batch / wall, "decode_scenes_per_s": batch / wall,
        "decode_weight_stream": {"achieved_GBs": gs["work"] / (gs["ms"] * 1e-3) / 1e9 if gs["ms"] > 0 else 0.0, "launches": int(gs["launches"])},
        "visible_fraction_of_causal_keys": visible_frac, "decode_path": path,
    }
    if path == "split":
        out["roofline_decode_attention"]["kernel"] = "ar_attn_kernel (decode_path = split: the attention-only kernel; q/k/v come from the LayerNorm + QKV projection kernel)"
    return out


# ----------------------------------------------------------------------------------------------------------------- config 5 across ranks
def config5_multi_rank(local_rank, rank, world, dist, steps):
    """BASELINE configs[4] across the ranks: every GPU decodes 64 sequences (16 BEV layouts x 4 samples; the samples of a layout stay on one rank - they share the
    condition prefix's K/V rows), top-k 32, explicit Philox-2025 uniforms, Route A config-4 model.  No data-path collective; one gather of the token ids to rank 0
    inside the timed region (bevgen_amd.parallel.gather_token_ids).  Timing as the contract says: barrier + synchronize on both sides, max over ranks."""
    import torch
    from bevgen_amd import presets, synthetic
    from bevgen_amd.parallel import gather_token_ids

    S, batch, top_k = 4, 64, 32
    cfg = presets.config4()
    if DRY_RUN:
        ctx = _StubContext(cfg)
        steps = steps or 8
    else:
        from bevgen_amd.runtime import Context
        from bevgen_amd.weights import gpt_state_dict
        ctx = Context(cfg, route="ar", device=local_rank, max_batch=batch, kv_cache="f32")
        ctx.load_state_dict(cached(("gpt", "config4"), lambda: gpt_state_dict(presets.config4(), 1234)))
        ctx.set_tables()
        ctx.finalize()
        steps = steps or cfg.num_img_tokens
    bt = synthetic.make_batch(cfg, batch // S, seed=5000 + rank)            # each rank: its own 16 layouts
    bt = {k: v.repeat_interleave(S, dim=0).to(ctx.device) for k, v in bt.items()}
    noise = synthetic.uniform_noise((steps, batch), 2025, 1 + rank).to(ctx.device)
    sync = (lambda: None) if DRY_RUN else torch.cuda.synchronize

    def run(n):
        ids = ctx.ar_sample(bt["cond_ids"], bt["intrinsics_inv"], bt["extrinsics_inv"], steps=n, greedy=False, top_k=top_k, temperature=1.0,
                            noise_u=noise[:n].contiguous() if n != steps else noise, samples_per_layout=S)
        return gather_token_ids(ids, dist)

    run(min(8, steps))   # warm-up (graph capture, workspace)
    if dist:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    out = run(steps)
    sync()
    mine = time.perf_counter() - t0
    if dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed, mine], dtype=torch.float64, device=ctx.device)
    per_rank = [t.clone() for _ in range(world)] if dist else [t]
    if dist:
        dist.all_gather(per_rank, t)
    elapsed = max(float(x[0]) for x in per_rank)
    ctx.close()
    if rank != 0:
        return None
    assert out is not None and out.shape[0] == world * batch
    return {"scaling": "weak", "sequences_per_gpu": batch, "layouts_per_gpu": batch // S, "samples_per_layout": S, "top_k": top_k, "decode_steps": steps,
            "sequences_per_s": world * batch / elapsed, "ms_per_decode_step": elapsed * 1e3 / steps, "per_rank_ms_per_decode_step": [float(x[1]) * 1e3 / steps for x in per_rank],
            "gathered_token_ids": list(out.shape),
            "config": "BASELINE configs[4]: Route A config-4 model, top-k 32, explicit uniforms (Philox seed 2025), 16 BEV layouts x 4 samples per GPU (a layout's samples share "
                      "one rank and its condition-prefix K/V), token ids gathered to rank 0 over RCCL"}


# ----------------------------------------------------------------------------------------------------------------- main
def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert DRY_RUN or torch.cuda.is_available(), "bench.py needs an MI355X"
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)")
    sync = (lambda: None) if DRY_RUN else torch.cuda.synchronize
    Event = _CpuEvent if DRY_RUN else torch.cuda.Event
    numa = None
    if not DRY_RUN:
        torch.cuda.set_device(local_rank)
        from bevgen_amd.parallel import bind_to_gpu_numa_node

        numa = bind_to_gpu_numa_node(local_rank)   # one rank = one GPU: keep its host thread on the GPU's NUMA node (launch jitter shows at 0.8 ms per replayed decode step)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist_mod.init_process_group("gloo" if DRY_RUN else "nccl", rank=rank, world_size=world)  # backend "nccl" is RCCL on ROCm
        dist = dist_mod
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"bench.py: RCCL world size {dist.get_world_size()} != --gpus {args.gpus}")
    n_gpus = dist.get_world_size() if dist else 1

    from bevgen_amd import synthetic
    from bevgen_amd.parallel import gather_scenes

    def run_route_m(precision, steps, warmup, cams, batch, weights="f32"):
        cfg, ctx, _ = build_route_m(cams, batch, local_rank, precision, weights)
        bt = synthetic.make_batch(cfg, batch, seed=1000 + rank)  # each rank: its own shard of scenes
        bt = {k: v.to(ctx.device) for k, v in bt.items()}
        ev = [[Event(enable_timing=True) for _ in range(4)] for _ in range(steps)]
        seeds = []

        def one_step(e=None):
            if e: e[0].record()
            # stochastic sampling like the reference's default generate (gumbel + critic noise, top-k 0.9): uniforms drawn inside the sampler kernels
            ids = ctx.maskgit_generate(bt["cond_ids"], bt["intrinsics_inv"], bt["extrinsics_inv"], timesteps=args.timesteps, noise_seed=2025 + 7919 * rank + len(seeds), check=False)
            seeds.append(0)
            if e: e[1].record()
            px = ctx.vq_decode(ids.reshape(batch * cams, -1), latent_hw=(cfg.cam_latent_h, cfg.cam_latent_w), uint8=True, check=False)      # [B*C,3,256,256] uint8
            # (check=False: the step stays asynchronous; the device status word - non-finite logits / pixels, f16 range - is read without a sync at the entry of the next
            # library call and, after the timed region's closing synchronisation, by ctx.synchronize() below: a flagged step fails the bench instead of being timed)
            if e: e[2].record()
            out = gather_scenes(px.reshape(batch, cams, 3, px.shape[-2], px.shape[-1]), dist)
            if e: e[3].record()
            return out

        for _ in range(warmup):
            one_step()
        if dist:
            dist.barrier()
        sync()
        # the timed region runs CLEAN: K steps, nothing but the product path and four stream events per step
        t0 = time.perf_counter()
        for i in range(steps):
            one_step(ev[i])
        sync()
        mine = time.perf_counter() - t0          # this rank's own K steps (before the closing barrier)
        if hasattr(ctx, "synchronize"):
            ctx.synchronize()                    # raises if any kernel of the K steps flagged its results
        if dist:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        # ... and the per-kernel roofline comes from a SEPARATE pass of the same step with a HIP-event pair around every GEMM / convolution / attention launch
        # (two extra hipEventRecords per launch would otherwise sit inside the headline number)
        prof_steps = min(steps, 2)
        ctx.profile_begin()
        tp0 = time.perf_counter()
        for i in range(prof_steps):
            one_step()
        sync()
        prof_elapsed = time.perf_counter() - tp0
        prof = ctx.profile_end()
        prof["_pass"] = {"steps": prof_steps, "ms": prof_elapsed * 1e3, "launches": 0}
        t = torch.tensor([elapsed, mine], dtype=torch.float64, device=ctx.device)
        per_rank = [t.clone() for _ in range(world)] if dist else [t]
        if dist:
            dist.all_gather(per_rank, t)
        parts = {"step": [e[0].elapsed_time(e[3]) for e in ev], "generate": [e[0].elapsed_time(e[1]) for e in ev],
                 "vq_decode": [e[1].elapsed_time(e[2]) for e in ev], "gather": [e[2].elapsed_time(e[3]) for e in ev],
                 "per_rank_ms_per_step": [float(x[1]) * 1e3 / steps for x in per_rank]}
        ctx.close()
        return max(float(x[0]) for x in per_rank), prof, parts, cfg

    elapsed, prof, parts, cfg_m = run_route_m(args.precision, args.steps, args.warmup, args.cams, args.batch)
    strong = None
    if world > 1 and 16 % world == 0:
        s_steps = max(2, min(args.steps, 5))
        e_s, _, p_s, _ = run_route_m(args.precision, s_steps, 1, args.cams, 16 // world)   # 16 scenes in total, sharded
        strong = {"scaling": "strong", "global_batch": 16, "scenes_per_gpu": 16 // world, "steps": s_steps, "value": 16 * s_steps / e_s, "unit": "scenes/s", "ms_per_step": e_s * 1e3 / s_steps,
                  "per_rank_ms_per_step": p_s["per_rank_ms_per_step"]}
    c5_multi = None
    if world > 1 and not args.no_decode_leg:
        c5_multi = config5_multi_rank(local_rank, rank, world, dist, args.decode_steps)
    exact = None
    if DRY_RUN:
        args.no_extra_legs = args.no_decode_leg = args.no_cpu_baseline = args.no_exact_leg = True
    if world == 1 and args.precision != "fp32" and not args.no_exact_leg:
        e2, p2, _, _ = run_route_m("fp32", 1, 1, args.cams, args.batch)   # the exact-fp32 parity mode on the same workload (one step)
        exact = (e2, p2)

    w16 = None
    if world == 1 and args.precision != "fp32" and not args.no_extra_legs:
        # the f16-weights model (GEMM / convolution matrices rounded to f16 at load, two MFMAs per product; tokens bit-exact vs the oracle on the rounded
        # weights, tests): NOT the headline - a different (rounded) model, reported beside it
        e4, p4, parts4, _ = run_route_m(args.precision, 2, 1, args.cams, args.batch, weights="f16")
        w16 = (e4, p4, parts4)

    if rank != 0:
        if dist:
            dist.destroy_process_group()
        return

    def roof(prof, precision, measure=False):
        g = prof["gemm"]
        ach = g["work"] / (g["ms"] * 1e-3) / 1e12 if g["ms"] > 0 else 0.0
        if precision == "fp32":
            peak, kern, note = MFMA_FP32_PEAK_TF, "gemm_f32_kernel<MODE_PLAIN>", "v_mfma_f32_32x32x2_f32, exact fp32"
        else:
            peak, kern, note = MFMA_F16_PEAK_TF / 3.0, "gemm_split_glds_kernel<MODE_PLAIN, 4, 3>", "3 v_mfma_f32_32x32x16_f16 per fp32 product (hi/lo split): ceiling = f16 dense peak / 3"
        r = {"bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak, "traffic": None, "kernel": kern, "note": note,
             "launches": int(g["launches"]), "avg_us": g["ms"] * 1e3 / max(g["launches"], 1),
             # launches of the same GEMM family that are not this kernel (small-problem blocks, the short last part of a row-split launch): timed apart, so that
             # launches / avg_us above are one kernel's and can be held against the rocprof trace
             "other_gemm_launches": {"launches": int(prof.get("gemm_small", {}).get("launches", 0)), "ms": prof.get("gemm_small", {}).get("ms", 0.0)}}
        if measure and world == 1 and precision != "fp32":
            tr = inrun_traffic("gemm_split_glds_kernel<0, 4, 3")
            if "error" in tr:
                r["traffic_note"] = "not measured: " + tr["error"]
            else:
                alg = gemm_algorithmic_bytes(cfg_m, args.batch)
                r["traffic"] = tr["bytes_per_launch"]
                r["traffic_detail"] = dict(tr, algorithmic_bytes_per_launch=alg, traffic_over_algorithmic=tr["bytes_per_launch"] / alg)
        return r

    import numpy as np
    scenes = n_gpus * args.batch * args.steps
    detail = {
        "metric": "multi-view scenes/sec (6x256x256)", "value": scenes / elapsed, "unit": "scenes/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed * 1e3 / args.steps, "ms_per_step_median": pct(parts["step"], 50), "ms_per_step_p99": pct(parts["step"], 99),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if args.precision == "fp32" else "f32 (GEMM/conv/attention products as 3 f16 MFMAs on hi/lo splits, fp32 accumulate)",
        "data": "synthetic" if not DRY_RUN else "DRY RUN: stub context on CPU, control flow only - no performance meaning",
        "config": {"workload": f"BASELINE configs[1]: Route M MaskGit, {args.cams}x256x256, batch {args.batch} scenes/GPU, 18 iterations, + VQGAN decode to uint8",
                   "global_batch": n_gpus * args.batch, "parallelism": f"scene-parallel x{n_gpus} (RCCL gather of uint8 pixels)", "precision_mode": args.precision},
        "rccl_world_size": n_gpus, "per_rank_ms_per_step": parts["per_rank_ms_per_step"], "host_numa_binding_rank0": numa,
        "ms_per_maskgit_iteration": float(np.mean(parts["generate"])) / args.timesteps,
        "vqgan_decode_ms_per_scene": float(np.mean(parts["vq_decode"])) / args.batch,
        "gather_ms_per_step": float(np.mean(parts["gather"])),
        "roofline": roof(prof, args.precision, measure=True),
        "kernel_time_share": {k: v["ms"] / prof["_pass"]["ms"] for k, v in prof.items() if v["launches"]},
        "kernel_tflops": {k: (v["work"] / (v["ms"] * 1e-3) / 1e12) for k, v in prof.items() if v["launches"] and k in ("gemm", "gemm_small", "conv3x3", "attention")},
        "profiled_pass": {"steps": prof["_pass"]["steps"], "ms_per_step": prof["_pass"]["ms"] / prof["_pass"]["steps"],
                          "note": "roofline / kernel_time_share / kernel_tflops come from this separate pass of the same step with a HIP-event pair around every hot launch; `value` is timed without it"},
    }
    line = detail
    if DRY_RUN:
        line["dry_run"] = True
    if strong is not None:
        line["strong_scaling"] = strong
    if c5_multi is not None:
        line["config5"] = c5_multi
    if exact is not None:
        e2, p2 = exact
        line["exact_fp32_mode"] = {"value": args.batch / e2, "unit": "scenes/s", "ms_per_step": e2 * 1e3, "roofline": roof(p2, "fp32"),
                                   "note": "bit-exact-parity mode (every product in fp32 on the matrix cores), same workload, 1 step"}
    if w16 is not None:
        e4, p4, parts4 = w16
        g4 = p4["gemm"]
        ach4 = g4["work"] / (g4["ms"] * 1e-3) / 1e12
        line["f16_weights_mode"] = {"value": args.batch * 2 / e4, "unit": "scenes/s", "ms_per_step": e4 * 1e3 / 2, "ms_per_maskgit_iteration": float(np.mean(parts4["generate"])) / args.timesteps,
                                    "vqgan_decode_ms_per_scene": float(np.mean(parts4["vq_decode"])) / args.batch,
                                    "roofline": {"bound": "mfma", "achieved": ach4, "peak": MFMA_F16_PEAK_TF / 2.0, "unit": "TFLOP/s", "frac": ach4 / (MFMA_F16_PEAK_TF / 2.0),
                                                 "kernel": "gemm_split_glds_kernel<MODE_PLAIN, 4, 3, W16>", "launches": int(g4["launches"]), "avg_us": g4["ms"] * 1e3 / max(g4["launches"], 1),
                                                 "note": "2 v_mfma_f32_32x32x16_f16 per product (activations hi + lo, weights one f16 plane): ceiling = f16 dense peak / 2"},
                                    "kernel_time_share": {k: v["ms"] / p4["_pass"]["ms"] for k, v in p4.items() if v["launches"]},
                                    "note": "Context(weights='f16'): every GEMM / convolution matrix rounded to f16 once at load (the reference's bf16 autocast rounds them to 8 bits on "
                                            "every call), activations and attention operands keep their hi + lo planes; same workload, 2 steps.  A different (rounded) model: the "
                                            "headline above is the fp32-weights model"}
    if world == 1 and not args.no_extra_legs:
        e3, _, p3, _ = run_route_m(args.precision, 2, 1, 3, args.batch)   # the shape of the released Argoverse checkpoint (3 cameras, N=768)
        line["released_3_camera_shape"] = {"value": args.batch * 2 / e3, "unit": "scenes/s", "ms_per_step": e3 * 1e3 / 2, "ms_per_maskgit_iteration": float(np.mean(p3["generate"])) / args.timesteps,
                                           "config": f"Route M, 3x256x256 (configs/modes/argoverse.yaml), batch {args.batch}, 2 steps"}
    if world == 1 and not args.no_extra_legs and args.batch > 1:
        # the interactive caller's shape (scripts/interactive_editing.py:273-277): ONE scene per call - latency, not throughput.  Same step (MaskGit generate + VQGAN decode
        # to uint8), the library picks its small-grid kernels by itself (64- / 128-row GEMM blocks, key-split self-attention, row-split launches)
        e1, _, p1, _ = run_route_m(args.precision, 3, 1, args.cams, 1)
        line["single_scene_latency"] = {"value": e1 * 1e3 / 3, "unit": "ms per scene", "higher_is_better": False, "ms_per_maskgit_iteration": float(np.mean(p1["generate"])) / args.timesteps,
                                        "config": f"Route M, {args.cams}x256x256, batch 1, 3 steps"}
    if world == 1 and (DRY_RUN or not args.no_extra_legs) and args.batch == 16:
        # No multi-GPU node has been available to builder or driver: the 16-scene STRONG-scaling leg at N GPUs puts 16 / N scenes on each GPU with no data-path
        # collective, so one GPU running batch 16 / N is that leg's per-GPU load.  prediction(N) = N x scenes/s at batch 16 / N (an upper bound: the gather of 19 MB
        # of uint8 pixels to rank 0 and launch skew come on top); N = 1 is the headline itself.
        proxy = {}
        for n in (2, 4, 8):
            ep, _, pp, _ = run_route_m(args.precision, 2, 1, args.cams, 16 // n)
            proxy[str(n)] = {"scenes_per_gpu": 16 // n, "single_gpu_scenes_per_s": (16 // n) * 2 / ep, "ms_per_step": ep * 1e3 / 2, "predicted_strong_scaling_scenes_per_s": n * (16 // n) * 2 / ep}
        line["strong_scaling_single_gpu_proxy"] = {"by_n_gpus": proxy, "weak_scaling_prediction_scenes_per_s": {str(n): n * scenes / elapsed for n in (2, 4, 8)},
                                                   "note": "measured on ONE GPU at the per-GPU batch of the N-GPU strong leg (16 scenes in total); scene-parallel with one final gather, so N x this is "
                                                           "the stated prediction until a node exists; weak scaling (16 scenes per GPU) predicts N x the headline"}
    dkeys = ("ms_per_decode_step", "ms_per_decode_step_median", "ms_per_decode_step_p99", "decode_scenes_per_s", "roofline_decode_attention", "decode_step_roofline", "visible_fraction_of_causal_keys")
    if world == 1 and not args.no_decode_leg:
        line.update(decode_leg(local_rank, args.decode_batch, args.decode_steps))
        # BASELINE config 4 names fp16 storage: the same decode with the KV cache stored as fp16 (fp32 accumulate; logits within 2e-3 of the range, tests)
        f16 = decode_leg(local_rank, args.decode_batch, args.decode_steps, kv_cache="f16")
        line["decode_f16_kv_cache"] = {k: f16[k] for k in dkeys}
        # ... and the all-fp16-storage model (projection weights rounded to fp16 at load, tokens bit-exact vs the oracle on the rounded weights; fp32 arithmetic)
        h16 = decode_leg(local_rank, args.decode_batch, args.decode_steps, kv_cache="f16", weights="f16")
        line["decode_f16_kv_cache_f16_weights"] = {k: h16[k] for k in dkeys}
        # ... and fp32 KV cache + fp16 weights: BIT-EXACT greedy tokens vs the oracle on the rounded weights (tests) with the 2-byte weight stream
        e16 = decode_leg(local_rank, args.decode_batch, args.decode_steps, kv_cache="f32", weights="f16")
        line["decode_f32_kv_cache_f16_weights"] = {k: e16[k] for k in dkeys}
        # the same kernel timed on the PRODUCT path (hipGraph replay) by a rocprofv3 --kernel-trace child pass: the cross-check of the HIP-event figure above
        if not DRY_RUN and os.environ.get("BEVGEN_BENCH_NO_PMC") != "1":
            try:
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import ktrace_inrun
                from bevgen_amd import presets as _presets
                steps_kt = args.decode_steps or _presets.config4().num_img_tokens
                kt = ktrace_inrun.measure("ar_attn_fused_kernel", args.decode_batch, steps_kt, "f16", "f16")
            except Exception as e:
                kt = {"error": f"{type(e).__name__}: {e}"}
            ra = line["decode_f16_kv_cache_f16_weights"]["roofline_decode_attention"]
            try:
                ra["traffic_inrun"] = ktrace_inrun.measure_traffic("ar_attn_fused_kernel", args.decode_batch, 16, "f16", "f16", timeout=150)
            except Exception as e:
                ra["traffic_inrun"] = {"error": f"{type(e).__name__}: {e}"}
            if "error" not in kt:
                cfg4 = _presets.config4()
                kvb, _ = route_a_bytes(cfg4, args.decode_batch, steps_kt, 2, 1, 2)
                per_launch = kvb / (steps_kt * cfg4.num_layers)
                gbs = per_launch / (kt["avg_us"] * 1e-6) / 1e9
                ra["kernel_trace"] = dict(kt, achieved=gbs, frac_by_survey_8d_fp16_bytes=gbs / HBM_PEAK_GBS, bytes_per_launch=per_launch,
                                          note="K/V bytes per launch (SURVEY 8d, mean over the decode) / average kernel duration in the replayed graph")
            else:
                ra["kernel_trace"] = kt
        if not args.no_extra_legs:
            # SURVEY 8(d) config 4 variant: density 0.35, every layer with its own random per-head block layouts - the key walk follows the chunk lists of present blocks
            d35 = decode_leg(local_rank, args.decode_batch, args.decode_steps, kv_cache="f16", density=0.35)
            line["decode_density_035_f16_kv_cache"] = {k: d35[k] for k in dkeys}
            d35h = decode_leg(local_rank, args.decode_batch, args.decode_steps, kv_cache="f16", weights="f16", density=0.35)
            line["decode_density_035_f16_kv_cache_f16_weights"] = {k: d35h[k] for k in dkeys}
            # the four-launch form: the decode-attention kernel proper (K/V stream only) with the projection as its own MFMA kernel, reported for its attention-kernel roofline
            sp = decode_leg(local_rank, args.decode_batch, args.decode_steps, kv_cache="f16", weights="f16", path="split")
            line["decode_split_path_f16_kv_cache_f16_weights"] = {k: sp[k] for k in dkeys}
            c5 = decode_leg(local_rank, 64, args.decode_steps, kv_cache="f32", S=4, top_k=32, stochastic=True)
            line["config5_topk32_4_samples_per_layout"] = {"sequences": 64, "layouts": 16, "ms_per_decode_step": c5["ms_per_decode_step"], "ms_per_decode_step_median": c5["ms_per_decode_step_median"],
                                                           "ms_per_decode_step_p99": c5["ms_per_decode_step_p99"], "sequences_per_s": c5["decode_sequences_per_s"], "prefill_ms": c5["decode_prefill_ms"],
                                                           "roofline_decode_attention": c5["roofline_decode_attention"], "decode_step_roofline": c5["decode_step_roofline"],
                                                           "config": "BASELINE configs[4] on one GPU: Route A config-4 model, top-k 32, explicit uniforms (Philox seed 2025), 16 BEV layouts x 4 samples, condition prefix prefilled and read once per layout"}
            # the same with the fp16 K/V cache + fp16 projection weights (BASELINE configs[3] names fp16; token ids then follow the near-tie criterion, not equality)
            c5h = decode_leg(local_rank, 64, args.decode_steps, kv_cache="f16", weights="f16", S=4, top_k=32, stochastic=True)
            line["config5_topk32_4_samples_per_layout_f16"] = {"sequences": 64, "layouts": 16, "ms_per_decode_step": c5h["ms_per_decode_step"], "ms_per_decode_step_median": c5h["ms_per_decode_step_median"],
                                                               "ms_per_decode_step_p99": c5h["ms_per_decode_step_p99"], "sequences_per_s": c5h["decode_sequences_per_s"],
                                                               "roofline_decode_attention": c5h["roofline_decode_attention"], "decode_step_roofline": c5h["decode_step_roofline"],
                                                               "config": "as config5_topk32_4_samples_per_layout, fp16 K/V cache + fp16 projection weights"}
    if world == 1 and not args.no_cpu_baseline and args.cpu_baseline != "none":
        line["cpu_baseline"] = cpu_baseline(args.cams, args.timesteps, args.cpu_baseline)

    # ---- what is printed: ONE short JSON line (contract keys + compact `roofline` / `cpu_baseline` + flat scalars per leg, `legs` LAST so that it survives a tail
    #      cut); everything above goes, unabridged, to the detail file next to it
    detail_path = os.path.join(ROOT, "gpurun_out", f"bench_detail_n{n_gpus}.json")
    try:
        os.makedirs(os.path.dirname(detail_path), exist_ok=True)
        json.dump(detail, open(detail_path, "w"), indent=1)
    except OSError as e:
        detail_path = f"not written: {e}"
    short = {k: detail[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                                    "rccl_world_size", "per_rank_ms_per_step")}
    if DRY_RUN:
        short["dry_run"] = True
    r = detail["roofline"]
    short["roofline"] = {k: r[k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "launches", "avg_us")}
    if r.get("traffic_detail"):
        short["roofline"]["traffic_over_algorithmic"] = r["traffic_detail"]["traffic_over_algorithmic"]
        short["roofline"]["traffic_source"] = "in-run rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child passes"
    elif r.get("traffic_note"):
        short["roofline"]["traffic_note"] = r["traffic_note"][:100]
    if "cpu_baseline" in detail:
        c = detail["cpu_baseline"]
        short["cpu_baseline"] = {"value": c["value"], "unit": c["unit"], "cores": c["cores"], "kind": c["kind"],
                                 "sample": f"1 scene end to end, {c['sample'].split('MaskGit generate with ')[1].split(' =')[0]} of {args.timesteps} scaled linearly, {c['host']['threads_used']} torch threads on {c['host']['cpu_model']}"[:118]}
    def rnd(x, n=4):
        return None if x is None else round(float(x), n)

    # the HBM roofline the north star names (decode attention, BASELINE config 4, fp16 storage), a first-class sibling of `roofline`
    hd = detail.get("decode_f16_kv_cache_f16_weights")
    if hd is not None:
        ra = hd["roofline_decode_attention"]
        kt = ra.get("kernel_trace") if isinstance(ra.get("kernel_trace"), dict) and "avg_us" in ra.get("kernel_trace", {}) else None
        tr = ra.get("traffic_inrun") or {}
        short["roofline_decode_attention"] = {
            "bound": "hbm", "kernel": "ar_attn_fused_kernel<kv=f16, G=1, w=f16> (ln1 + qkv + decode attention: K/V bytes / WHOLE kernel time)",
            "achieved": rnd(ra["achieved"], 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": rnd(ra["frac"]), "launches": ra["launches"], "avg_us": rnd(ra["avg_us"], 2),
            "traffic": tr.get("bytes_per_launch"), "traffic_over_algorithmic": rnd(tr.get("traffic_over_algorithmic")), 
            "trace_avg_us": rnd(kt["avg_us"], 2) if kt else None, "trace_frac": rnd(kt["frac_by_survey_8d_fp16_bytes"]) if kt else None,
            "workload": "BASELINE configs[3]: Route A config 4, B=16, all contexts of the 2100-token decode; achieved = SURVEY 8(d) bytes / per-launch HIP events; trace_* = rocprofv3 "
                        "kernel trace of the replayed graph; traffic = in-run PMC at short contexts"}
        if "error" in tr:
            short["roofline_decode_attention"]["traffic_note"] = str(tr["error"])[:100]
    elif DRY_RUN:
        short["roofline_decode_attention"] = {"bound": "hbm", "kernel": "ar_attn_fused_kernel<kv=f16, G=1, w=f16>", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None,
                                              "launches": 0, "avg_us": None, "traffic": None}
    short["detail_file"] = os.path.relpath(detail_path, ROOT) if os.path.isabs(detail_path) else detail_path

    def dshort(d):
        ra, rs = d["roofline_decode_attention"], d["decode_step_roofline"]
        return {"ms_step": rnd(d["ms_per_decode_step"]), "median": rnd(d["ms_per_decode_step_median"]), "p99": rnd(d["ms_per_decode_step_p99"]),
                "attn_frac_8d_fp16_bytes": rnd(ra.get("frac_by_survey_8d_fp16_bytes")), "attn_frac_storage_bytes": rnd(ra["frac"]), "attn_frac_kv_plus_qkv_w": rnd(ra.get("frac_kv_plus_qkv_weight_bytes")),
                "attn_avg_us": rnd(ra["avg_us"], 2), "attn_walk_phase_us": rnd(ra["attention_phase"]["us"], 2), "step_frac": rnd(rs["frac"]),
                **({"attn_trace_avg_us": rnd(ra["kernel_trace"]["avg_us"], 2), "attn_trace_frac_8d_fp16_bytes": rnd(ra["kernel_trace"]["frac_by_survey_8d_fp16_bytes"])}
                   if isinstance(ra.get("kernel_trace"), dict) and "avg_us" in ra["kernel_trace"] else {})}

    legs = {"ms_per_step_median": rnd(detail["ms_per_step_median"], 2), "ms_per_step_p99": rnd(detail["ms_per_step_p99"], 2), "ms_per_maskgit_iteration": rnd(detail["ms_per_maskgit_iteration"], 3),
            "vqgan_decode_ms_per_scene": rnd(detail["vqgan_decode_ms_per_scene"], 3), "gather_ms_per_step": rnd(detail["gather_ms_per_step"], 3),
            "kernel_tflops": {k: rnd(v, 1) for k, v in detail["kernel_tflops"].items()}, "kernel_time_share": {k: rnd(v, 3) for k, v in detail["kernel_time_share"].items() if k != "_pass"}}
    if strong is not None:
        legs["strong_scaling"] = {"global_batch": 16, "scenes_per_s": rnd(strong["value"]), "ms_per_step": rnd(strong["ms_per_step"], 2), "per_rank_ms_per_step": [rnd(x, 2) for x in strong["per_rank_ms_per_step"]]}
    if c5_multi is not None:
        legs["config5"] = {"sequences_per_s": rnd(c5_multi["sequences_per_s"], 2), "ms_per_decode_step": rnd(c5_multi["ms_per_decode_step"]), "sequences_per_gpu": 64, "decode_steps": c5_multi["decode_steps"],
                           "per_rank_ms_per_decode_step": [rnd(x) for x in c5_multi["per_rank_ms_per_decode_step"]]}
    if "strong_scaling_single_gpu_proxy" in detail:
        px = detail["strong_scaling_single_gpu_proxy"]
        legs["scaling_prediction_from_one_gpu"] = {"strong_16_scenes": {n: rnd(v["predicted_strong_scaling_scenes_per_s"], 2) for n, v in px["by_n_gpus"].items()},
                                                   "per_gpu_scenes_per_s_at_batch": {str(v["scenes_per_gpu"]): rnd(v["single_gpu_scenes_per_s"], 3) for v in px["by_n_gpus"].values()},
                                                   "weak_16_per_gpu": {n: rnd(v, 2) for n, v in px["weak_scaling_prediction_scenes_per_s"].items()}}
    if "single_scene_latency" in detail:
        legs["single_scene_latency_ms"] = rnd(detail["single_scene_latency"]["value"], 2)
    if "released_3_camera_shape" in detail:
        legs["released_3_camera_scenes_per_s"] = rnd(detail["released_3_camera_shape"]["value"], 3)
    if "exact_fp32_mode" in detail:
        legs["exact_fp32"] = {"scenes_per_s": rnd(detail["exact_fp32_mode"]["value"], 3), "gemm_frac": rnd(detail["exact_fp32_mode"]["roofline"]["frac"], 3)}
    if "f16_weights_mode" in detail:
        legs["f16_weights"] = {"scenes_per_s": rnd(detail["f16_weights_mode"]["value"], 3), "gemm_frac": rnd(detail["f16_weights_mode"]["roofline"]["frac"], 3)}
    if "ms_per_decode_step" in detail:
        # (the driver keeps the LAST ~2000 characters of the output: the legs the north star is judged on come last, and `roofline_decode_attention` after `legs`)
        if "config5_topk32_4_samples_per_layout" in detail:
            c = detail["config5_topk32_4_samples_per_layout"]
            legs["config5_64seq_1gpu"] = {"ms_step": rnd(c["ms_per_decode_step"]), "median": rnd(c["ms_per_decode_step_median"]), "p99": rnd(c["ms_per_decode_step_p99"]),
                                          "sequences_per_s": rnd(c["sequences_per_s"], 2), "attn_frac_storage_bytes": rnd(c["roofline_decode_attention"]["frac"]), "step_frac": rnd(c["decode_step_roofline"]["frac"])}
        if "config5_topk32_4_samples_per_layout_f16" in detail:
            c = detail["config5_topk32_4_samples_per_layout_f16"]
            legs["config5_64seq_1gpu_f16"] = {"ms_step": rnd(c["ms_per_decode_step"]), "sequences_per_s": rnd(c["sequences_per_s"], 2), "attn_frac_storage_bytes": rnd(c["roofline_decode_attention"]["frac"]),
                                              "step_frac": rnd(c["decode_step_roofline"]["frac"])}
        dl = {"prefill_ms": rnd(detail["decode_prefill_ms"], 2)}
        for name, key in (("split_path_f16_kv_f16_w", "decode_split_path_f16_kv_cache_f16_weights"), ("density035_f16_kv", "decode_density_035_f16_kv_cache"),
                          ("density035_f16_kv_f16_w", "decode_density_035_f16_kv_cache_f16_weights"), ("f16_kv", "decode_f16_kv_cache"), ("f32_kv", None),
                          ("f32_kv_f16_w", "decode_f32_kv_cache_f16_weights"), ("f16_kv_f16_w", "decode_f16_kv_cache_f16_weights")):
            if key is None:
                dl[name] = dshort(detail)
            elif key in detail:
                dl[name] = dshort(detail[key])
        legs["decode_config4_B16"] = dl
    short["legs"] = legs
    if "roofline_decode_attention" in short:
        short["roofline_decode_attention"] = short.pop("roofline_decode_attention")   # last key of the line
    print(json.dumps(short), flush=True)
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
