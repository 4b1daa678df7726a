#!/usr/bin/env python
"""Launch the two rooflined kernels a few times with bench-sized operands (for `rocprofv3 --pmc` passes: FETCH_SIZE / WRITE_SIZE).
    gemm:    M=24576 (16 scenes x 1536 tokens), N=1024, K=1024 fp32  -> algorithmic bytes = (M*K + N*K + M*N)*4
    decode:  B=16, H=16, n=1500 of Lmax=2368, fp32 KV               -> algorithmic bytes = 2*B*H*n*64*4
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bevgen_amd.runtime import Context

ctx = Context(None)
M, N, K = 24576, 1024, 1024
a = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda")
for _ in range(5):
    ctx.op_gemm(a, w)
# split-precision LDS-DMA GEMM (the production path of the default f16x3 mode): operands are split on the fly by the op entry point,
# the GEMM kernel itself reads 4 B per element of A and B (interleaved hi/lo f16 planes) and writes fp32 C -> same algorithmic bytes
from bevgen_amd.runtime import _ptr, _stream
out = torch.empty(M, N, device="cuda")
for _ in range(5):
    ctx._check(ctx.lib.bevgen_op_gemm(ctx._h, _ptr(a), _ptr(w), None, None, _ptr(out), M, N, K, 0, 3, _stream()))
B, H, n, L = 16, 16, 1500, 2368
q = torch.randn(B, H * 64, device="cuda")
kc = torch.randn(B, H, L, 64, device="cuda"); vc = torch.randn(B, H, L, 64, device="cuda")
bias = torch.randn(L, L, device="cuda")
for _ in range(5):
    ctx.op_decode_attention(q, kc, vc, n, bias=bias)
torch.cuda.synchronize()
print("gemm algorithmic MB", (M * K + N * K + M * N) * 4 / 1e6, "decode algorithmic MB", 2 * B * H * n * 64 * 4 / 1e6)
