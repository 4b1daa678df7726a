#!/usr/bin/env python
"""Phase stamps of the LDS-DMA GEMM's workgroups (build with -DBEVGEN_GEMM_TRACE, see tools/gemm_trace.sh): where a round of tiles spends its time.
usage: BEVGEN_LIB_PATH=.../libbevgen_hip_trace.so python tools/gemm_trace.py M,N,K [residual]"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bevgen_amd.runtime import Context, _ptr, _stream

M, N, K = (int(v) for v in sys.argv[1].split(","))
res = len(sys.argv) > 2
ctx = Context(None)
a = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.03
out = torch.zeros(M, N, device="cuda")
def run():
    ctx._check(ctx.lib.bevgen_op_gemm(ctx._h, _ptr(a), _ptr(w), None, _ptr(out) if res else None, _ptr(out), M, N, K, 0, 3, _stream()))
for _ in range(3): run()
torch.cuda.synchronize()
buf = np.zeros(2048 * 8, dtype=np.uint64)
lib = ctypes.CDLL(os.environ["BEVGEN_LIB_PATH"])
lib.bevgen_debug_gemm_trace.argtypes = [ctypes.c_void_p]
rc = lib.bevgen_debug_gemm_trace(buf.ctypes.data)
assert rc == 0, rc
t = buf.reshape(2048, 8).astype(np.int64)
n = min(2048, ((M + 255) // 256) * ((N + 127) // 128))
t = t[:n]
t0 = t[:, 0].min()
us = lambda x: x * 0.01   # 100 MHz ticks
print(f"M={M} N={N} K={K} residual={res}: {n} workgroups, span {us(t[:, 4].max() - t0):.1f} us")
cu = (t[:, 5] >> 32) * 4096 + ((t[:, 5] & 0xFFFF) >> 8)
print(f"distinct (xcc, se/sh/cu) ids: {len(set(cu.tolist()))}")
for r in range((n + 255) // 256):
    s = t[r * 256:(r + 1) * 256]
    f = lambda v: f"{us(np.median(v)):6.2f} [{us(v.min()):6.2f} .. {us(v.max()):6.2f}]"
    print(f"round {r}: entry at {f(s[:, 0] - t0)} us | fill {f(s[:, 1] - s[:, 0])} | loop {f(s[:, 2] - s[:, 1])} | epilogue issue {f(s[:, 3] - s[:, 2])} | drain {f(s[:, 4] - s[:, 3])} | end at {f(s[:, 4] - t0)}")
# turnaround on a CU: end of a workgroup -> entry of the next one on the same (xcc, cu)
gaps = []
for c in set(cu.tolist()):
    idx = np.where(cu == c)[0]
    idx = idx[np.argsort(t[idx, 0])]
    for i, j in zip(idx, idx[1:]): gaps.append(t[j, 0] - t[i, 4])
if gaps:
    g = np.array(gaps)
    print(f"turnaround (stores acknowledged -> next workgroup's first instruction on that CU): median {us(np.median(g)):.2f} us, min {us(g.min()):.2f}, max {us(g.max()):.2f}  ({len(g)} pairs)")
