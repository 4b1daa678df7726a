#!/usr/bin/env python
"""Phase stamps of the LDS-DMA GEMM's workgroups (build with -DBEVGEN_GEMM_TRACE, see tools/gemm_trace.sh): where a round of tiles spends its time.
usage: BEVGEN_LIB_PATH=.../libbevgen_hip_trace.so python tools/gemm_trace.py M,N,K [residual]   (an operator launch)
       BEVGEN_LIB_PATH=... python tools/gemm_trace.py model BATCH                                    (one Route-M step: the LAST throughput launch of every epilogue kind)"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bevgen_amd.runtime import Context, _ptr, _stream

KINDS = ["plain", "muse_q", "geglu", "muse_kv", "muse_qkv"]
lib = ctypes.CDLL(os.environ["BEVGEN_LIB_PATH"])
lib.bevgen_debug_gemm_trace.argtypes = [ctypes.c_void_p]
us = lambda x: x * 0.01   # 100 MHz ticks

def report(t, label):
    gx, gy = int(t[0, 6] >> 32), int(t[0, 6] & 0xFFFFFFFF)
    n = min(2048, gx * gy)
    if n == 0: return
    t = t[:n]
    flags = int(t[0, 7] & 0xFFFFFFFF)
    t0 = t[:, 0].min()
    print(f"{label}: grid {gx} x {gy}, K={int(t[0, 7] >> 32)}, residual={flags & 1} ln consumer={(flags >> 1) & 1} ln producer={(flags >> 2) & 1}: {n} workgroups traced, span {us(t[:, 4].max() - t0):.1f} us")
    for r in range(min(3, (n + 255) // 256)):
        s = t[r * 256:(r + 1) * 256]
        f = lambda v: f"{us(np.median(v)):6.2f} [{us(v.min()):6.2f} .. {us(v.max()):6.2f}]"
        print(f"  round {r}: entry at {f(s[:, 0] - t0)} us | fill {f(s[:, 1] - s[:, 0])} | loop {f(s[:, 2] - s[:, 1])} | epilogue issue {f(s[:, 3] - s[:, 2])} | drain {f(s[:, 4] - s[:, 3])}")
    cu = (t[:, 5] >> 32) * 4096 + ((t[:, 5] & 0xFFFF) >> 8)
    gaps = []
    for c in set(cu.tolist()):
        idx = np.where(cu == c)[0]
        idx = idx[np.argsort(t[idx, 0])]
        for i, j in zip(idx, idx[1:]): gaps.append(t[j, 0] - t[i, 4])
    if gaps:
        g = np.array(gaps)
        print(f"  turnaround on a CU (stores acknowledged -> next workgroup's first instruction): median {us(np.median(g)):.2f} us, min {us(g.min()):.2f}, max {us(g.max()):.2f}")

def read():
    buf = np.zeros(5 * 2048 * 8, dtype=np.uint64)
    rc = lib.bevgen_debug_gemm_trace(buf.ctypes.data)
    assert rc == 0, rc
    return buf.reshape(5, 2048, 8).astype(np.int64)

if sys.argv[1] == "model":
    import bench
    from bevgen_amd import synthetic
    B = int(sys.argv[2])
    cfg, ctx, _ = bench.build_route_m(6, B, 0, "f16x3", "f32")
    bt = {k: v.to(ctx.device) for k, v in synthetic.make_batch(cfg, B, seed=1000).items()}
    ctx.maskgit_generate(bt["cond_ids"], bt["intrinsics_inv"], bt["extrinsics_inv"], timesteps=2, noise_seed=2025, check=False)
    torch.cuda.synchronize(); ctx.synchronize()
    t = read()
    for k in range(5): report(t[k], KINDS[k])
else:
    M, N, K = (int(v) for v in sys.argv[1].split(","))
    res = len(sys.argv) > 2
    ctx = Context(None)
    a = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.03
    out = torch.zeros(M, N, device="cuda")
    for _ in range(3):
        ctx._check(ctx.lib.bevgen_op_gemm(ctx._h, _ptr(a), _ptr(w), None, _ptr(out) if res else None, _ptr(out), M, N, K, 0, 3, _stream()))
    torch.cuda.synchronize()
    report(read()[0], f"M={M} N={N} K={K}")
