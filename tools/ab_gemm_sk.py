"""Per-shape timing of the stream-K GEMM against the launcher's other choice (bevgen_op_gemm mode 6 vs 3), HIP events over 200 launches.  usage: python tools/ab_gemm_sk.py"""
import sys, math, torch
sys.path.insert(0, '.')
from bevgen_amd.runtime import Context, _ptr, _stream
ctx = Context(None, precision="f16x3")
for (M, N, K) in [(1536, 1024, 1024), (1536, 3072, 1024), (1536, 5504, 1024), (1536, 1024, 2752), (3072, 1024, 1024), (3072, 3072, 1024), (3072, 5504, 1024), (3072, 1024, 2752),
                  (6144, 3072, 1024), (6144, 5504, 1024), (6144, 1024, 2752)]:
    a = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") / math.sqrt(K); out = torch.empty(M, N, device="cuda")
    res = {}
    for mode in (3, 6):
        for _ in range(5): ctx._check(ctx.lib.bevgen_op_gemm(ctx._h, _ptr(a), _ptr(w), None, None, _ptr(out), M, N, K, 0, mode, _stream()))
        torch.cuda.synchronize()
        # (the op entry re-splits both operands per call: time the pair of split kernels alone and subtract)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100): ctx._check(ctx.lib.bevgen_op_gemm(ctx._h, _ptr(a), _ptr(w), None, None, _ptr(out), M, N, K, 0, mode, _stream()))
        e1.record(); torch.cuda.synchronize()
        res[mode] = e0.elapsed_time(e1) * 10.0
    print(f"M={M} N={N} K={K}: per call (incl. the operand split kernels of the op entry) tiles-per-workgroup {res[3]:.1f} us, stream-K {res[6]:.1f} us, delta {res[6]-res[3]:+.1f} us")
ctx.synchronize()
