"""Operator-level parity: each HIP kernel (called through the C ABI) vs a plain PyTorch-CPU fp32 computation of the same op."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def rel(a, b):
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def dev(t):
    return t.cuda().contiguous()


@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (200, 136, 64), (1, 7, 32), (777, 1024, 1024), (300, 5460, 1024), (260, 1024, 2752)])
@pytest.mark.parametrize("epi", ["none", "bias_gelu_res"])
def test_gemm(gpu_ctx, M, N, K, epi):
    g = torch.Generator().manual_seed(M * 7 + N)
    a = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    # asymmetric operands (a transposed output or swapped operand cannot pass)
    if epi == "none":
        ref = a.double() @ w.double().t()
        out = gpu_ctx.op_gemm(dev(a), dev(w))
    else:
        b = torch.randn(N, generator=g)
        r = torch.randn(M, N, generator=g)
        ref = F.gelu((a.double() @ w.double().t()) + b.double()) + r.double()
        out = gpu_ctx.op_gemm(dev(a), dev(w), dev(b), dev(r), gelu=True)
    assert rel(out.cpu().double(), ref) < 6e-6  # fp32 accumulate: ~sqrt(K) eps


@pytest.mark.parametrize("M,N,K", [(1, 64, 128), (16, 3072, 1024), (33, 1024, 4096), (64, 1000, 1024), (48, 4096, 1024)])
def test_gemm_skinny(gpu_ctx, M, N, K):
    g = torch.Generator().manual_seed(M + N)
    a = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    b = torch.randn(N, generator=g)
    r = torch.randn(M, N, generator=g)
    ref = F.gelu((a.double() @ w.double().t()) + b.double()) + r.double()
    out = gpu_ctx.op_gemm(dev(a), dev(w), dev(b), dev(r), gelu=True, skinny=True)
    assert rel(out.cpu().double(), ref) < 6e-6  # fp32 accumulate: ~sqrt(K) eps
    out2 = gpu_ctx.op_gemm(dev(a), dev(w), skinny=True)
    assert rel(out2.cpu().double(), a.double() @ w.double().t()) < 6e-6


@pytest.mark.parametrize("rows,D,beta", [(5, 128, True), (1000, 1024, False), (37, 1024, True)])
def test_layernorm(gpu_ctx, rows, D, beta):
    g = torch.Generator().manual_seed(rows)
    x = torch.randn(rows, D, generator=g) * 3 + 1
    gamma = torch.randn(D, generator=g)
    b = torch.randn(D, generator=g) if beta else None
    ref = F.layer_norm(x, (D,), gamma, b, 1e-5)
    out = gpu_ctx.op_layernorm(dev(x), dev(gamma), dev(b) if beta else None)
    assert rel(out.cpu(), ref) < 1e-5


@pytest.mark.parametrize("rows,Fi", [(9, 341), (130, 2730)])
def test_geglu_layernorm(gpu_ctx, rows, Fi):
    g = torch.Generator().manual_seed(Fi)
    h = torch.randn(rows, 2 * Fi, generator=g)
    gamma = torch.randn(Fi, generator=g)
    a, gate = h.chunk(2, dim=-1)
    ref = F.layer_norm(gate * F.gelu(a), (Fi,), gamma, None, 1e-5)
    ldy = (Fi + 31) // 32 * 32
    out = gpu_ctx.op_geglu_layernorm(dev(h), dev(gamma), ldy=ldy).cpu()
    assert rel(out[:, :Fi], ref) < 1e-5
    assert out[:, Fi:].abs().max() == 0  # zero padding consumed by the following GEMM


# (the bias-less form exists for key counts that are multiples of 32: only those shapes are generated for it)
@pytest.mark.parametrize("B,H,Nq,Nk,use_bias", [(*sh, ub) for ub in (True, False) for sh in [(1, 2, 48, 49), (2, 3, 200, 257), (1, 16, 300, 17), (2, 2, 128, 128)] if ub or sh[3] % 32 == 0])
def test_attention(gpu_ctx, B, H, Nq, Nk, use_bias):
    g = torch.Generator().manual_seed(Nq + Nk)
    q = torch.randn(B, H, Nq, 64, generator=g)
    k = torch.randn(B, H, Nk, 64, generator=g)
    v = torch.randn(B, H, Nk, 64, generator=g)
    Nk_pad = (Nk + 31) // 32 * 32
    kp = torch.zeros(B, H, Nk_pad, 64); kp[:, :, :Nk] = k
    vp = torch.zeros(B, H, Nk_pad, 64); vp[:, :, :Nk] = v
    scale = 0.31
    sim = torch.einsum("bhid,bhjd->bhij", q.double(), k.double()) * scale
    bias = None
    if use_bias:
        bias_real = torch.randn(Nq, Nk, generator=g) * 2
        bias_real[::3, 1::2] = -1e30  # masked entries
        bias_real[:, 0] = 0.5         # never a fully masked row
        sim = sim + bias_real.double()
        bias = torch.full((Nq, Nk_pad), -1e30)
        bias[:, :Nk] = bias_real
    ref = torch.einsum("bhij,bhjd->bhid", sim.softmax(-1), v.double()).permute(0, 2, 1, 3).reshape(B, Nq, H * 64)
    out = gpu_ctx.op_attention(dev(q), dev(kp), dev(vp), dev(bias) if use_bias else None, scale)
    assert rel(out.cpu().double(), ref) < 5e-6


@pytest.fixture(scope="module")
def gpu_ctx_split():
    """A model-less context in split-precision mode: bevgen_op_attention then runs the Route M flash-attention kernel (8-wave ping-pong, attention_split.hip)."""
    from bevgen_amd.runtime import Context

    ctx = Context(None, precision="f16x3")
    yield ctx
    ctx.close()


@pytest.mark.parametrize("B,H,Nq,Nk,use_bias", [(*sh, ub) for ub in (True, False)
                                                for sh in [(1, 2, 48, 49), (2, 3, 200, 257), (1, 16, 300, 17), (2, 2, 128, 128), (1, 2, 513, 1568), (2, 1, 256, 32)] if ub or sh[3] % 32 == 0])
def test_attention_split_precision(gpu_ctx_split, B, H, Nq, Nk, use_bias):
    """The split-precision kernel at operator level against fp64: ragged query counts (not multiples of 32 / 256), one-tile and 49-tile key ranges, masked
    entries, and the bias-less form (the launcher's zero block)."""
    g = torch.Generator().manual_seed(Nq + Nk)
    q = torch.randn(B, H, Nq, 64, generator=g)
    k = torch.randn(B, H, Nk, 64, generator=g)
    v = torch.randn(B, H, Nk, 64, generator=g)
    Nk_pad = (Nk + 31) // 32 * 32
    kp = torch.zeros(B, H, Nk_pad, 64); kp[:, :, :Nk] = k
    vp = torch.zeros(B, H, Nk_pad, 64); vp[:, :, :Nk] = v
    scale = 0.31
    sim = torch.einsum("bhid,bhjd->bhij", q.double(), k.double()) * scale
    bias = None
    if use_bias:
        bias_real = torch.randn(Nq, Nk, generator=g) * 2
        bias_real[::3, 1::2] = -1e30
        bias_real[:, 0] = 0.5
        sim = sim + bias_real.double()
        bias = torch.full((Nq, Nk_pad), -1e30)
        bias[:, :Nk] = bias_real
    ref = torch.einsum("bhij,bhjd->bhid", sim.softmax(-1), v.double()).permute(0, 2, 1, 3).reshape(B, Nq, H * 64)
    out = gpu_ctx_split.op_attention(dev(q), dev(kp), dev(vp), dev(bias) if use_bias else None, scale)
    assert rel(out.cpu().double(), ref) < 5e-6


@pytest.mark.parametrize("B,H,Nq,Nk,splits", [(1, 2, 300, 1537, 2), (2, 3, 77, 250, 3), (1, 1, 256, 64, 2), (1, 4, 513, 1568, 5)])
def test_attention_split_precision_key_ranges(gpu_ctx_split, B, H, Nq, Nk, splits):
    """Key-split form of the split-precision kernel (the low-latency self-attention of Route M at one scene per call): the key tiles cut into `splits` ranges - uneven
    ones, one-tile ones, a range whose keys are all masked for some rows, the spike of the rescale branch in the last range - merged by the combine kernel; fp64 reference,
    same bound as the unsplit kernel, and bit-identical run to run."""
    g = torch.Generator().manual_seed(Nq + Nk + splits)
    q = torch.randn(B, H, Nq, 64, generator=g)
    k = torch.randn(B, H, Nk, 64, generator=g)
    v = torch.randn(B, H, Nk, 64, generator=g)
    k[0, 0, Nk - 3] = q[0, 0, 5] * 2.5                      # late spike: the last range holds the row maximum
    Nk_pad = (Nk + 31) // 32 * 32
    kp = torch.zeros(B, H, Nk_pad, 64); kp[:, :, :Nk] = k
    vp = torch.zeros(B, H, Nk_pad, 64); vp[:, :, :Nk] = v
    scale = 0.31
    bias_real = torch.randn(Nq, Nk, generator=g) * 2
    bias_real[::3, 1::2] = -1e30
    bias_real[1::4, : Nk // 2] = -1e30                        # rows that see nothing of the first range(s)
    bias_real[:, Nk - 1] = 0.5
    sim = torch.einsum("bhid,bhjd->bhij", q.double(), k.double()) * scale + bias_real.double()
    bias = torch.full((Nq, Nk_pad), -1e30)
    bias[:, :Nk] = bias_real
    ref = torch.einsum("bhij,bhjd->bhid", sim.softmax(-1), v.double()).permute(0, 2, 1, 3).reshape(B, Nq, H * 64)
    dq, dk, dv, db = dev(q), dev(kp), dev(vp), dev(bias)
    out = gpu_ctx_split.op_attention(dq, dk, dv, db, scale, key_splits=splits).cpu()
    one = gpu_ctx_split.op_attention(dq, dk, dv, db, scale).cpu()
    assert rel(out.double(), ref) < 5e-6
    assert rel(out.double(), one.double()) < 2e-6
    assert torch.equal(out, gpu_ctx_split.op_attention(dq, dk, dv, db, scale, key_splits=splits).cpu())


def test_attention_split_precision_rescale_branch(gpu_ctx_split):
    """A key far above the running maximum in a LATE tile (every accumulator rescaled exactly once), split-precision kernel."""
    B, H, Nq, Nk = 1, 1, 64, 256
    g = torch.Generator().manual_seed(5)
    q = torch.randn(B, H, Nq, 64, generator=g)
    k = torch.randn(B, H, Nk, 64, generator=g) * 0.1
    v = torch.randn(B, H, Nk, 64, generator=g)
    k[0, 0, 200] = q[0, 0, 7] * 3.0
    k[0, 0, 40] = q[0, 0, 9] * 2.0
    sim = torch.einsum("bhid,bhjd->bhij", q.double(), k.double())
    ref = torch.einsum("bhij,bhjd->bhid", sim.softmax(-1), v.double()).permute(0, 2, 1, 3).reshape(B, Nq, 64)
    out = gpu_ctx_split.op_attention(dev(q), dev(k), dev(v), None, 1.0)
    assert rel(out.cpu().double(), ref) < 5e-6


def test_attention_online_softmax_rescale_branch(gpu_ctx):
    """A key far above the running maximum appears in a LATE tile: every accumulator must be rescaled exactly once."""
    B, H, Nq, Nk = 1, 1, 64, 256
    g = torch.Generator().manual_seed(5)
    q = torch.randn(B, H, Nq, 64, generator=g)
    k = torch.randn(B, H, Nk, 64, generator=g) * 0.1
    v = torch.randn(B, H, Nk, 64, generator=g)
    k[0, 0, 200] = q[0, 0, 7] * 3.0   # spike for query 7 in tile 6
    k[0, 0, 40] = q[0, 0, 9] * 2.0    # and an early one for query 9
    sim = torch.einsum("bhid,bhjd->bhij", q.double(), k.double())
    ref = torch.einsum("bhij,bhjd->bhid", sim.softmax(-1), v.double()).permute(0, 2, 1, 3).reshape(B, Nq, 64)
    out = gpu_ctx.op_attention(dev(q), dev(k), dev(v), None, 1.0)
    assert rel(out.cpu().double(), ref) < 5e-6


@pytest.mark.parametrize("B,H,n,Lmax", [(1, 2, 1, 64), (2, 4, 77, 128), (3, 16, 600, 640), (16, 16, 2368, 2368)])
@pytest.mark.parametrize("masked", [False, True])
def test_decode_attention(gpu_ctx, B, H, n, Lmax, masked):
    g = torch.Generator().manual_seed(n)
    q = torch.randn(B, H * 64, generator=g)
    kc = torch.randn(B, H, Lmax, 64, generator=g)
    vc = torch.randn(B, H, Lmax, 64, generator=g)
    L = Lmax
    bias = torch.randn(L, L, generator=g)
    keep = None
    scale = 0.125
    qh = q.reshape(B, H, 64).double()
    s = (torch.einsum("bhd,bhjd->bhj", qh, kc[:, :, :n].double()) + bias[n - 1, :n].double()) * scale
    if masked:
        keep = (torch.rand(H, L, L, generator=g) > 0.3).to(torch.uint8)
        keep[:, :, 0] = 1
        s = s.masked_fill(keep[:, n - 1, :n][None] == 0, float("-inf"))
    ref = torch.einsum("bhj,bhjd->bhd", s.softmax(-1), vc[:, :, :n].double()).reshape(B, H * 64)
    out = gpu_ctx.op_decode_attention(dev(q), dev(kc), dev(vc), n, bias=dev(bias), keep=dev(keep) if masked else None, scale=scale)
    assert rel(out.cpu().double(), ref) < 5e-6


@pytest.mark.parametrize("n,H,W,Cin,Cout,up", [(2, 8, 8, 64, 128, False), (1, 16, 16, 32, 32, True), (3, 5, 7, 32, 3, False), (1, 32, 32, 128, 64, False)])
def test_conv3x3(gpu_ctx, n, H, W, Cin, Cout, up):
    g = torch.Generator().manual_seed(H * W)
    x = torch.randn(n, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)
    b = torch.randn(Cout, generator=g)
    xin = F.interpolate(x, scale_factor=2.0, mode="nearest") if up else x
    ref = F.conv2d(xin.double(), w.double(), b.double(), padding=1)
    res = torch.randn_like(ref, dtype=torch.float32)
    ref = ref + res.double()
    out = gpu_ctx.op_conv3x3(dev(x.permute(0, 2, 3, 1)), dev(w.permute(0, 2, 3, 1)), dev(b), residual=dev(res.permute(0, 2, 3, 1)), upsample=up)
    assert rel(out.cpu().permute(0, 3, 1, 2).double(), ref) < 2e-6


@pytest.mark.parametrize("n,H,W,Cin,Cout", [(3, 5, 7, 32, 3),          # fewer rows than one tile, ragged everything, three output channels
                                            (2, 16, 16, 64, 128),      # whole 128-row tiles, alone on their CUs (the eight-wave block)
                                            (1, 14, 25, 128, 96),      # the non-square latent grid of the 224 x 400 cameras
                                            (1, 200, 200, 32, 64),     # 313 blocks of 128 rows (two per CU)
                                            (2, 128, 128, 32, 256)])   # 256 blocks of 256 rows: the throughput shape
def test_conv3x3_lds_dma_kernel(gpu_ctx, n, H, W, Cin, Cout):
    """The LDS-DMA convolution at operator level (the decoder reaches it with planes its GroupNorm wrote): against fp64, and the stride-1 variant (wave-uniform tap
    displacements + a per-lane tap mask, MODE_CONV3S) bit-identical to the general one on every block shape the launcher picks."""
    g = torch.Generator().manual_seed(H * W + Cin)
    x = torch.randn(n, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)
    b = torch.randn(Cout, generator=g)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    res = torch.randn_like(ref, dtype=torch.float32)
    ref = ref + res.double()
    args = (dev(x.permute(0, 2, 3, 1)), dev(w.permute(0, 2, 3, 1)), dev(b))
    fast = gpu_ctx.op_conv3x3(*args, residual=dev(res.permute(0, 2, 3, 1)), kernel="dma")
    general = gpu_ctx.op_conv3x3(*args, residual=dev(res.permute(0, 2, 3, 1)), kernel="dma_general")
    assert rel(fast.cpu().permute(0, 3, 1, 2).double(), ref) < 2e-6
    assert torch.equal(fast, general)


@pytest.mark.parametrize("n,hw,C,swish", [(2, 64, 32, True), (1, 4096, 128, True), (3, 256, 512, False), (2, 100, 64, True)])
def test_groupnorm(gpu_ctx, n, hw, C, swish):
    g = torch.Generator().manual_seed(C)
    h = int(math.isqrt(hw))
    x = torch.randn(n, C, h, hw // h, generator=g) * 2 + 0.7
    gamma = torch.randn(C, generator=g)
    beta = torch.randn(C, generator=g)
    ref = F.group_norm(x, 32, gamma, beta, 1e-6)
    if swish:
        ref = ref * torch.sigmoid(ref)
    out = gpu_ctx.op_groupnorm(dev(x.permute(0, 2, 3, 1)), dev(gamma), dev(beta), swish=swish)
    assert rel(out.cpu().permute(0, 3, 1, 2), ref) < 1e-5


@pytest.mark.parametrize("mode", [2, 3, 4])  # 2: register-staged kernel (fp32 A split on the fly), 3: LDS-DMA kernel (pre-split A planes), 4: the same with an
@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (200, 136, 64), (777, 1024, 1024), (300, 5460, 1024), (260, 1024, 2752)])   # f16-representable weight: 2 MFMAs / product
def test_gemm_split_precision(gpu_ctx, M, N, K, mode):
    """3x f16 MFMA on (hi, lo*2^-11) splits: fp32-class accuracy (a handful of fp32 ulps beyond the exact fp32 path)."""
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g) * 3.0          # LayerNorm-like magnitudes
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    if mode == 4:
        w = w.half().float()                          # the weights='f16' mode rounds its matrices once; the kernel then skips the (zero) low plane
    b = torch.randn(N, generator=g)
    r = torch.randn(M, N, generator=g)
    ref = F.gelu((a.double() @ w.double().t()) + b.double()) + r.double()
    out = torch.empty(M, N, device="cuda")
    from bevgen_amd.runtime import _ptr, _stream
    da, dw, db, dr = dev(a), dev(w), dev(b), dev(r)  # keep the device tensors alive across the raw-pointer call
    gpu_ctx._check(gpu_ctx.lib.bevgen_op_gemm(gpu_ctx._h, _ptr(da), _ptr(dw), _ptr(db), _ptr(dr), _ptr(out), M, N, K, 1, mode, _stream()))
    err = rel(out.cpu().double(), ref)
    exact = rel(gpu_ctx.op_gemm(dev(a), dev(w), dev(b), dev(r), gelu=True).cpu().double(), ref)
    assert err < 2e-6, (err, exact)


@pytest.mark.parametrize("mode", [2, 3])
def test_gemm_split_precision_small_and_large_magnitudes(gpu_ctx, mode):
    g = torch.Generator().manual_seed(0)
    a = torch.randn(256, 256, generator=g)
    a[:, ::7] *= 1e-5     # tiny entries (f16-subnormal hi parts) must not lose the result
    a[:, 3::11] *= 3e3    # large but in-range
    w = torch.randn(128, 256, generator=g) * 0.05
    ref = a.double() @ w.double().t()
    out = torch.empty(256, 128, device="cuda")
    from bevgen_amd.runtime import _ptr, _stream
    da, dw = dev(a), dev(w)
    gpu_ctx._check(gpu_ctx.lib.bevgen_op_gemm(gpu_ctx._h, _ptr(da), _ptr(dw), None, None, _ptr(out), 256, 128, 256, 0, mode, _stream()))
    assert rel(out.cpu().double(), ref) < 2e-6


# ----------------------------------------------------------------------------------------------------------------- fused Route A decode kernels (operator level)
def _ar_attn_reference(x, partial, rbias, ln_w, ln_b, wqkv, bqkv, kc, vc, n, bias, mask, layout, blk, G, prefix, scale=0.125):
    """fp64 statement of one decode row through ln1 -> q/k/v -> cache append -> masked softmax attention -> + ln1(x) (gpt:240-253, ssa:150-176).
    Returns (x2 [B, D], new k row [B, H, 64], new v row).  Groups of G sequences read keys [0, prefix) from the group's first cache slot."""
    B, D = x.shape
    H = D // 64
    row = x.double()
    if partial is not None:
        row = row + (partial.double().sum(0) + (rbias.double() if rbias is not None else 0))
    xn = F.layer_norm(row, (D,), ln_w.double(), ln_b.double(), 1e-5)
    qkv = xn @ wqkv.double().t() + bqkv.double()
    q, k, v = (t.reshape(B, H, 64) for t in qkv.split(D, dim=1))
    kcd, vcd = kc.double().clone(), vc.double().clone()
    kcd[:, :, n - 1], vcd[:, :, n - 1] = k, v
    out = torch.empty(B, H, 64, dtype=torch.float64)
    r = n - 1
    for b in range(B):
        src = torch.full((n,), b)
        if G > 1:
            src[: min(prefix, n)] = (b // G) * G
        kk = kcd[src, :, torch.arange(n)].permute(1, 0, 2)   # [H, n, 64]
        vv = vcd[src, :, torch.arange(n)].permute(1, 0, 2)
        s = torch.einsum("hd,hjd->hj", q[b], kk)
        if bias is not None:
            s = s + bias[r, :n].double()
        s = s * scale
        vis = torch.ones(H, n, dtype=torch.bool)
        if mask is not None:
            vis &= (mask[r, :n] != 0)[None]
        if layout is not None:
            vis &= layout[:, r // blk, torch.arange(n) // blk] != 0
        s = s.masked_fill(~vis, float("-inf"))
        out[b] = torch.einsum("hj,hjd->hd", s.softmax(-1), vv)
    return xn + out.reshape(B, D), k, v


@pytest.mark.parametrize("split", [False, True])
@pytest.mark.parametrize("kv", ["f32", "f16"])
@pytest.mark.parametrize("B,G,H,n,Lmax,blk,sparse", [
    (1, 1, 2, 1, 64, 16, False),        # first key only
    (2, 1, 4, 77, 128, 4, True),        # ragged length, 4-key blocks (several blocks per 16-key chunk)
    (3, 1, 16, 600, 640, 16, True),
    (16, 1, 16, 2368, 2368, 16, True),  # BASELINE config 4 at the last position, n = L
    (16, 1, 16, 1301, 2368, 16, False),
    (4, 2, 4, 300, 512, 16, True),      # two sequences per layout sharing a 64-key prefix
    (8, 4, 8, 333, 512, 32, True),      # four per layout, 32-key blocks (two chunks per block)
    (8, 4, 8, 64, 512, 16, False),      # nothing beyond the shared prefix yet except the new key at n - 1 = 63 ... prefix 48
])
def test_ar_attn_fused_operator(gpu_ctx, B, G, H, n, Lmax, blk, sparse, kv, split):
    """ar_attn_fused_kernel (ln1 + q/k/v + cache append + block-sparse decode attention + residual) against fp64, over batch sizes, context lengths incl. n = 1 and
    n = L, layout groups G in {1, 2, 4}, both cache dtypes, with element mask + random per-head block layout (absent blocks are skipped, not masked).
    split: the four-launch form (LayerNorm + QKV projection kernel with the row source folded into its A fetch, then ar_attn_kernel)."""
    D = H * 64
    g = torch.Generator().manual_seed(B * 1000 + n)
    x = torch.randn(B, D, generator=g)
    partial = torch.randn(3, B, D, generator=g) * 0.3
    rbias = torch.randn(D, generator=g) * 0.1
    ln_w, ln_b = torch.randn(D, generator=g) * 0.2 + 1, torch.randn(D, generator=g) * 0.1
    wqkv = torch.randn(3 * D, D, generator=g) / math.sqrt(D)
    bqkv = torch.randn(3 * D, generator=g) * 0.1
    kc = torch.randn(B, H, Lmax, 64, generator=g)
    vc = torch.randn(B, H, Lmax, 64, generator=g)
    bias = torch.randn(Lmax, Lmax, generator=g)
    prefix = 0 if G == 1 else (48 if n <= 64 else 64)
    mask = layout = None
    if sparse:
        mask = (torch.rand(Lmax, Lmax, generator=g) > 0.2).float()
        layout = (torch.rand(H, Lmax // blk, Lmax // blk, generator=g) < 0.4).long()
        mask[:, 0] = 1
        layout[:, :, 0] = 1   # every row keeps something
    if kv == "f16":
        kc, vc = kc.half().float(), vc.half().float()   # the cache holds fp16-representable values; the appended row is rounded by the kernel
    ref, k_new, v_new = _ar_attn_reference(x, partial, rbias, ln_w, ln_b, wqkv, bqkv, kc, vc, n, bias, mask, layout, blk, G, prefix)
    cdt = torch.float16 if kv == "f16" else torch.float32
    dkc, dvc = dev(kc.to(cdt)), dev(vc.to(cdt))
    out = gpu_ctx.op_ar_attn_fused(dev(x), dev(ln_w), dev(ln_b), dev(wqkv), dev(bqkv), dkc, dvc, n, partial=dev(partial), rbias=dev(rbias), bias=dev(bias),
                                   attn_mask=None if mask is None else dev(mask), layout=None if layout is None else dev(layout), block=blk, G=G, prefix=prefix,
                                   kv_dtype=1 if kv == "f16" else 0, split=split)
    tol = 2e-3 if kv == "f16" else 2e-5   # f16: the new k/v row is stored rounded (11 bits) before it is read back
    assert rel(out.cpu().double(), ref) < tol
    # the appended rows
    assert rel(dkc[:, :, n - 1].float().cpu().double(), k_new) < (1e-3 if kv == "f16" else 1e-5)
    assert rel(dvc[:, :, n - 1].float().cpu().double(), v_new) < (1e-3 if kv == "f16" else 1e-5)
    # nothing else in the cache was touched
    keep = torch.ones(Lmax, dtype=torch.bool)
    keep[n - 1] = False
    assert torch.equal(dkc[:, :, keep].float().cpu(), kc[:, :, keep]) and torch.equal(dvc[:, :, keep].float().cpu(), vc[:, :, keep])


@pytest.mark.parametrize("form", [0, -1])
@pytest.mark.parametrize("kv", ["f32", "f16"])
@pytest.mark.parametrize("B,H,n,Lmax,masked,wf16", [
    (2, 4, 1, 64, False, False), (2, 4, 2, 64, True, False),             # no / one key in the cache
    (3, 8, 128, 256, True, False), (3, 8, 129, 256, False, True),        # n - 1 = 127 / 128: below / exactly one pipeline step of the 8 walking waves
    (2, 16, 257, 2368, True, False), (2, 16, 385, 512, False, False),    # whole steps + a ragged rest
    (2, 16, 256, 512, False, False), (2, 8, 513, 1024, True, False), (2, 8, 600, 1024, False, True),
    (16, 16, 1301, 2368, True, True),                                    # BASELINE config 4 mid-decode: 4 (fp16) / 2 (fp32) staged steps, the rest from HBM
    (16, 16, 2368, 2368, False, False),                                  # n = L
    (5, 2, 700, 1024, True, False),
])
def test_ar_attn_fused2_operator(gpu_ctx, B, H, n, Lmax, masked, wf16, kv, form):
    """The fused decode kernel for one sequence per workgroup and a dense walk against fp64, with the leading K/V steps of every wave's key walk staged in LDS by LDS-DMA
    (form 0 = as the sampling path launches it; -1 = no staging).  Context lengths around the staging boundaries (a pipeline step of the 16 waves = 256 keys), n = 1 and
    n = L, element mask (camera-bias visibility) on / off incl. a hidden new key, fp32 / fp16 weights (fp16: ln1 folded into the projection), split-K partial sums + bias
    folded into the row fetch; the appended rows and "nothing else touched".  (The block-sparse walk with staged chunks is covered by test_ar_attn_fused_operator.)"""
    D = H * 64
    g = torch.Generator().manual_seed(B * 1000 + n)
    x = torch.randn(B, D, generator=g)
    partial = torch.randn(4, B, D, generator=g) * 0.3
    rbias = torch.randn(D, generator=g) * 0.1
    ln_w, ln_b = torch.randn(D, generator=g) * 0.2 + 1, torch.randn(D, generator=g) * 0.1
    wqkv = torch.randn(3 * D, D, generator=g) / math.sqrt(D)
    bqkv = torch.randn(3 * D, generator=g) * 0.1
    kc = torch.randn(B, H, Lmax, 64, generator=g)
    vc = torch.randn(B, H, Lmax, 64, generator=g)
    bias = torch.randn(Lmax, Lmax, generator=g)
    mask = None
    if masked:
        mask = (torch.rand(Lmax, Lmax, generator=g) > 0.2).float()
        mask[:, 0] = 1
        mask[torch.arange(Lmax), torch.arange(Lmax)] = 1 if n % 2 else 0   # the new key itself visible / hidden
    if kv == "f16":
        kc, vc = kc.half().float(), vc.half().float()
    ref, k_new, v_new = _ar_attn_reference(x, partial, rbias, ln_w, ln_b, wqkv.half().float() if wf16 else wqkv, bqkv, kc, vc, n, bias, mask, None, 16, 1, 0)
    cdt = torch.float16 if kv == "f16" else torch.float32
    dkc, dvc = dev(kc.to(cdt)), dev(vc.to(cdt))
    out = gpu_ctx.op_ar_attn_fused(dev(x), dev(ln_w), dev(ln_b), dev(wqkv), dev(bqkv), dkc, dvc, n, partial=dev(partial), rbias=dev(rbias), bias=dev(bias),
                                   attn_mask=None if mask is None else dev(mask), kv_dtype=1 if kv == "f16" else 0, w_f16=wf16, split=form)
    assert rel(out.cpu().double(), ref) < (2e-3 if kv == "f16" else 2e-5)
    assert rel(dkc[:, :, n - 1].float().cpu().double(), k_new) < (1e-3 if kv == "f16" else 1e-5)
    assert rel(dvc[:, :, n - 1].float().cpu().double(), v_new) < (1e-3 if kv == "f16" else 1e-5)
    keep = torch.ones(Lmax, dtype=torch.bool)
    keep[n - 1] = False
    assert torch.equal(dkc[:, :, keep].float().cpu(), kc[:, :, keep]) and torch.equal(dvc[:, :, keep].float().cpu(), vc[:, :, keep])


@pytest.mark.parametrize("split", [False, True])
def test_ar_attn_fused_operator_f16_weights(gpu_ctx, split):
    B, H, n, Lmax = 16, 16, 700, 1024
    D = H * 64
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, D, generator=g)
    ln_w, ln_b = torch.ones(D), torch.zeros(D)
    wqkv = torch.randn(3 * D, D, generator=g) / math.sqrt(D)
    bqkv = torch.randn(3 * D, generator=g) * 0.1
    kc, vc = torch.randn(B, H, Lmax, 64, generator=g), torch.randn(B, H, Lmax, 64, generator=g)
    ref, _, _ = _ar_attn_reference(x, None, None, ln_w, ln_b, wqkv.half().float(), bqkv, kc, vc, n, None, None, None, 16, 1, 0)
    out = gpu_ctx.op_ar_attn_fused(dev(x), dev(ln_w), dev(ln_b), dev(wqkv), dev(bqkv), dev(kc), dev(vc), n, w_f16=True, split=split)
    assert rel(out.cpu().double(), ref) < 2e-5


@pytest.mark.parametrize("kv", ["f32", "f16"])
@pytest.mark.parametrize("B,H,n,Lmax,sparse,ks", [(1, 16, 2368, 2368, False, 4), (1, 16, 1301, 2368, True, 4), (2, 16, 700, 1024, False, 2), (1, 4, 40, 512, False, 4),
                                                   (1, 4, 1, 128, False, 3), (3, 8, 511, 512, True, 2)])
def test_ar_attn_key_split_operator(gpu_ctx, B, H, n, Lmax, sparse, ks, kv):
    """The attention-only kernel with its key walk cut into ks ranges (one- and two-sequence calls: more workgroups than (sequence, head) pairs) + the combine
    kernel, against fp64: long and short contexts (fewer chunks than ranges, n = 1), dense interleaved walk (range ends inside a 256-key span) and chunk lists."""
    D = H * 64
    g = torch.Generator().manual_seed(B * 77 + n)
    x = torch.randn(B, D, generator=g)
    ln_w, ln_b = torch.randn(D, generator=g) * 0.2 + 1, torch.randn(D, generator=g) * 0.1
    wqkv = torch.randn(3 * D, D, generator=g) / math.sqrt(D)
    bqkv = torch.randn(3 * D, generator=g) * 0.1
    kc, vc = torch.randn(B, H, Lmax, 64, generator=g), torch.randn(B, H, Lmax, 64, generator=g)
    bias = torch.randn(Lmax, Lmax, generator=g)
    mask = layout = None
    if sparse:
        mask = (torch.rand(Lmax, Lmax, generator=g) > 0.2).float()
        layout = (torch.rand(H, Lmax // 16, Lmax // 16, generator=g) < 0.4).long()
        mask[:, 0] = 1
        layout[:, :, 0] = 1
    if kv == "f16":
        kc, vc = kc.half().float(), vc.half().float()
    ref, k_new, _ = _ar_attn_reference(x, None, None, ln_w, ln_b, wqkv, bqkv, kc, vc, n, bias, mask, layout, 16, 1, 0)
    cdt = torch.float16 if kv == "f16" else torch.float32
    dkc, dvc = dev(kc.to(cdt)), dev(vc.to(cdt))
    out = gpu_ctx.op_ar_attn_fused(dev(x), dev(ln_w), dev(ln_b), dev(wqkv), dev(bqkv), dkc, dvc, n, bias=dev(bias), attn_mask=None if mask is None else dev(mask),
                                   layout=None if layout is None else dev(layout), block=16, kv_dtype=1 if kv == "f16" else 0, split=ks)
    assert rel(out.cpu().double(), ref) < (2e-3 if kv == "f16" else 2e-5)
    assert rel(dkc[:, :, n - 1].float().cpu().double(), k_new) < (1e-3 if kv == "f16" else 1e-5)


@pytest.mark.parametrize("M,N,K,ln,gelu", [(16, 4096, 1024, True, True), (16, 1024, 4096, False, False), (64, 1024, 1024, True, False), (5, 1000, 1024, True, False),
                                           (16, 256, 256, True, True), (33, 512, 2048, False, False)])
def test_ln_gemm_operator(gpu_ctx, M, N, K, ln, gelu):
    """skinny_fused_kernel through bevgen_op_ln_gemm: act(LayerNorm?(A) W^T + bias) for the decode-step shapes (ln2 + MLP-up + GELU, MLP-down split over K with the
    partial sums added by the consumer, ln_f + head) against fp64."""
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g) * 1.7 + 0.3
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    b = torch.randn(N, generator=g) * 0.2
    lw, lb = torch.randn(K, generator=g) * 0.2 + 1, torch.randn(K, generator=g) * 0.1
    h = F.layer_norm(a.double(), (K,), lw.double(), lb.double(), 1e-5) if ln else a.double()
    ref = h @ w.double().t()
    if ln:   # the split-K form carries neither bias nor activation (the consumer adds them)
        ref = ref + b.double()
        if gelu:
            ref = F.gelu(ref)
    out = gpu_ctx.op_ln_gemm(dev(a), dev(w), ln_w=dev(lw) if ln else None, ln_b=dev(lb) if ln else None, bias=dev(b) if ln else None, gelu=gelu and ln)
    assert rel(out.cpu().double(), ref) < 6e-6
    if ln:   # the folded-LayerNorm form the decode step launches (row statistics in the shadow of the product, applied in the epilogue)
        out = gpu_ctx.op_ln_gemm(dev(a), dev(w), ln_w=dev(lw), ln_b=dev(lb), bias=dev(b), gelu=gelu, ksplit=-1)
        assert rel(out.cpu().double(), ref) < 6e-6


@pytest.mark.parametrize("M,D,w_f16", [(16, 1024, False), (16, 1024, True), (5, 1024, True), (1, 1024, False), (9, 1024, False), (64, 1024, True), (33, 1024, False)])
def test_mlp_fused_operator(gpu_ctx, M, D, w_f16):
    """ar_mlp_fused_kernel through bevgen_op_mlp_fused: Linear2(GELU(Linear1(LayerNorm(x)))) of Block.forward (transformer/mingpt_sparse.py:232-253) in ONE launch - the
    up-projection's 16-column tiles exchanged inside each XCD, the down-projection as 8 partial planes summed by the consumer's row source, rows in chunks of 16 (M <= 64:
    BASELINE config 5's 64 sequences) - against fp64; the entry
    launches it twice on one barrier state (self-cleaning, as the hipGraph replay needs it) and fails on a barrier timeout or a workgroup off its XCD."""
    g = torch.Generator().manual_seed(7 * M + D + int(w_f16))
    x = torch.randn(M, D, generator=g) * 1.3 + 0.2
    lw, lb = torch.randn(D, generator=g) * 0.2 + 1, torch.randn(D, generator=g) * 0.1
    w1 = torch.randn(4 * D, D, generator=g) / math.sqrt(D)
    b1 = torch.randn(4 * D, generator=g) * 0.2
    w2 = torch.randn(D, 4 * D, generator=g) / math.sqrt(4 * D)
    b2 = torch.randn(D, generator=g) * 0.2
    w1r, w2r = (w1.half().float(), w2.half().float()) if w_f16 else (w1, w2)
    h = F.layer_norm(x.double(), (D,), lw.double(), lb.double(), 1e-5)
    ref = F.gelu(h @ w1r.double().t() + b1.double()) @ w2r.double().t() + b2.double()
    out = gpu_ctx.op_mlp_fused(dev(x), dev(lw), dev(lb), dev(w1), dev(b1), dev(w2), dev(b2), w_f16=w_f16)
    assert rel(out.cpu().double(), ref) < 8e-6
    out2 = gpu_ctx.op_mlp_fused(dev(x), dev(lw), dev(lb), dev(w1), dev(b1), dev(w2), dev(b2), w_f16=w_f16)
    assert torch.equal(out, out2), "two launches of the same operator differ: the partial planes must be summed in a fixed order"
    with pytest.raises(Exception, match="unsupported shape"):
        gpu_ctx.op_mlp_fused(dev(torch.randn(65, D)), dev(lw), dev(lb), dev(w1), dev(b1), dev(w2), dev(b2))
    with pytest.raises(Exception, match="unsupported shape"):   # other widths run the two skinny launches in the model path
        gpu_ctx.op_mlp_fused(dev(torch.randn(4, 512)), dev(lw[:512]), dev(lb[:512]), dev(w1[:2048, :512]), dev(b1[:2048]), dev(w2[:512, :2048]), dev(b2[:512]))


@pytest.mark.parametrize("scale", [1e-3, 1.0, 40.0])
def test_mlp_fused_operator_f16_matrix_pipe_over_input_magnitudes(gpu_ctx, scale):
    """fp16 weight storage: the launch multiplies on the f16 matrix pipe with the activation split hi + lo 2^-11 (two v_mfma_f32_16x16x32_f16 per 32 k).  The split keeps
    22 mantissa bits whatever the magnitude (lo is scaled out of the subnormal range), so rows whose LayerNorm gain / bias put them three decades apart - and hidden values
    from ~0 to a few hundred behind the GELU - must meet the same fp64 bound as the unit-scale case."""
    M, D = 16, 1024
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(M, D, generator=g)
    lw, lb = (torch.randn(D, generator=g) * 0.2 + 1) * scale, torch.randn(D, generator=g) * 0.1 * scale   # ln2's output carries the scale
    w1 = (torch.randn(4 * D, D, generator=g) / math.sqrt(D)).half().float()
    b1 = torch.randn(4 * D, generator=g) * 0.2 * scale
    w2 = (torch.randn(D, 4 * D, generator=g) / math.sqrt(4 * D)).half().float()
    b2 = torch.randn(D, generator=g) * 0.2
    h = F.layer_norm(x.double(), (D,), lw.double(), lb.double(), 1e-5)
    ref = F.gelu(h @ w1.double().t() + b1.double()) @ w2.double().t() + b2.double()
    out = gpu_ctx.op_mlp_fused(dev(x), dev(lw), dev(lb), dev(w1), dev(b1), dev(w2), dev(b2), w_f16=True)
    assert rel(out.cpu().double(), ref) < 8e-6


def test_profiler_state_is_per_context(gpu_ctx):
    """bevgen_profile_begin / _end time the launches of THEIR context only: another context's calls in between are not recorded (and do not disturb the records)."""
    from bevgen_amd.runtime import Context

    other = Context(None)
    a, w = dev(torch.randn(256, 128)), dev(torch.randn(128, 128))
    gpu_ctx.profile_begin()
    gpu_ctx.op_gemm(a, w)
    other.op_gemm(a, w)          # not profiled: `other` never called profile_begin
    other.profile_begin()
    other.op_gemm(a, w)
    gpu_ctx.op_gemm(a, w)
    po = other.profile_end()
    pa = gpu_ctx.profile_end()
    assert pa["gemm"]["launches"] == 2 and po["gemm"]["launches"] == 1, (pa["gemm"], po["gemm"])
    assert pa["gemm"]["ms"] > 0 and po["gemm"]["ms"] > 0
    other.close()


@pytest.mark.parametrize("M,N,K", [(1536, 1024, 1024), (768, 1024, 2752), (200, 136, 352)])
def test_gemm_split_precision_split_k(gpu_ctx, M, N, K):
    """Low-latency path of small batches: the LDS-DMA GEMM with its k range cut into three slices on gridDim.z (partial tiles added in slice order by splitk_reduce,
    then bias / GELU / residual) - same accuracy class as the unsplit kernel, and bit-identical run to run."""
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g) * 3.0
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    b = torch.randn(N, generator=g)
    r = torch.randn(M, N, generator=g)
    ref = F.gelu((a.double() @ w.double().t()) + b.double()) + r.double()
    from bevgen_amd.runtime import _ptr, _stream
    da, dw, db, dr = dev(a), dev(w), dev(b), dev(r)
    outs = []
    for _ in range(2):
        out = torch.empty(M, N, device="cuda")
        gpu_ctx._check(gpu_ctx.lib.bevgen_op_gemm(gpu_ctx._h, _ptr(da), _ptr(dw), _ptr(db), _ptr(dr), _ptr(out), M, N, K, 1, 5, _stream()))
        outs.append(out.cpu())
    assert rel(outs[0].double(), ref) < 2e-6
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("M,N,K", [(1536, 1024, 1024), (1536, 3072, 1024), (1536, 1024, 2752), (1536, 5504, 1024), (3072, 1024, 1024), (300, 384, 192), (1000, 136, 64), (256, 128, 32)])
def test_gemm_split_precision_stream_k(gpu_ctx, M, N, K):
    """The stream-K form of the LDS-DMA GEMM (gemm_split_glds_sk_kernel): units of one k-tile of one 256 x 128 tile dealt out evenly over the CUs, partial tiles published
    through the workspace and added by the workgroup that holds a tile's last k range, fused epilogue there.  The one- and two-scene Route-M shapes (q|k|v, the 1024-wide
    projections, the down- and up-projection), ragged shapes, and a problem smaller than one tile: same accuracy class as the one-workgroup-per-tile launch, bit-identical
    run to run, equal to that launch up to fp32 association."""
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g) * 3.0
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    b = torch.randn(N, generator=g)
    r = torch.randn(M, N, generator=g)
    ref = F.gelu((a.double() @ w.double().t()) + b.double()) + r.double()
    from bevgen_amd.runtime import _ptr, _stream
    da, dw, db, dr = dev(a), dev(w), dev(b), dev(r)
    outs = {}
    for mode in (6, 6, 3):
        out = torch.empty(M, N, device="cuda")
        gpu_ctx._check(gpu_ctx.lib.bevgen_op_gemm(gpu_ctx._h, _ptr(da), _ptr(dw), _ptr(db), _ptr(dr), _ptr(out), M, N, K, 1, mode, _stream()))
        outs.setdefault(mode, []).append(out.cpu())
    gpu_ctx.synchronize()
    assert rel(outs[6][0].double(), ref) < 2e-6
    assert torch.equal(outs[6][0], outs[6][1])
    assert rel(outs[6][0].double(), outs[3][0].double()) < 1e-6


@pytest.mark.parametrize("shape", ["2", "8", "16"])
def test_gemm_small_problem_block_shapes(shape):
    """The LDS-DMA GEMM picks its small-problem block by the grid size (eight waves with 32x64 patches on a four-stage ring when every block has a CU to itself,
    64-row blocks of four waves when even those cover at most half of the CUs, four waves on two stages otherwise).  $BEVGEN_GEMM_STAGES pins one of them (read once per process, hence the subprocess): both must give the fp64 product on the
    same shapes - ragged M and N, one k-tile, fewer k-tiles than ring stages, split-K, the f16-weights form."""
    import subprocess, sys, os
    code = r'''
import math, torch, sys
sys.path.insert(0, %r)
from bevgen_amd.runtime import Context, _ptr, _stream
ctx = Context(None)
worst = 0.0
for (M, N, K) in [(128, 128, 32), (200, 136, 64), (333, 264, 96), (777, 1024, 1024), (1536, 1024, 2752), (3072, 1024, 1024)]:
    for mode in (3, 4, 5):
        if mode == 5 and K // 32 < 6:
            continue
        g = torch.Generator().manual_seed(M + N + K)
        a = torch.randn(M, K, generator=g) * 3.0
        w = torch.randn(N, K, generator=g) / math.sqrt(K)
        if mode == 4:
            w = w.half().float()
        b = torch.randn(N, generator=g); r = torch.randn(M, N, generator=g)
        ref = torch.nn.functional.gelu(a.double() @ w.double().t() + b.double()) + r.double()
        da, dw, db, dr = a.cuda(), w.cuda(), b.cuda(), r.cuda()
        out = torch.empty(M, N, device="cuda")
        ctx._check(ctx.lib.bevgen_op_gemm(ctx._h, _ptr(da), _ptr(dw), _ptr(db), _ptr(dr), _ptr(out), M, N, K, 1, mode, _stream()))
        err = float((out.cpu().double() - ref).norm() / ref.norm())
        worst = max(worst, err)
        assert err < 2e-6, (M, N, K, mode, err)
print("worst", worst)
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BEVGEN_GEMM_STAGES=shape)
    res = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    assert "worst" in res.stdout


def test_gemm_row_split_launches_are_bit_identical():
    """A problem whose last round of 256 x 128 tiles would be nearly empty is launched as two row ranges (whole rounds + a short round of 128-row blocks).  Same kernels,
    same per-row arithmetic: the result must equal the single launch bit for bit ($BEVGEN_GEMM_ROWSPLIT=0, read once per process, hence two subprocesses) and the fp64
    product within the split-precision bound - ragged M, bias + GELU + residual epilogue, the f16-weights form."""
    import subprocess, sys, os, hashlib
    code = r'''
import math, torch, sys, hashlib
sys.path.insert(0, %r)
from bevgen_amd.runtime import Context, _ptr, _stream
ctx = Context(None)
for (M, N, K, mode) in [(8448 + 37, 1024, 256, 3), (66 * 256, 1024, 128, 4), (3072, 5504, 64, 3)]:
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g) * 3.0
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    if mode == 4:
        w = w.half().float()
    b = torch.randn(N, generator=g); r = torch.randn(M, N, generator=g)
    ref = torch.nn.functional.gelu(a.double() @ w.double().t() + b.double()) + r.double()
    da, dw, db, dr = a.cuda(), w.cuda(), b.cuda(), r.cuda()
    out = torch.full((M, N), float("nan"), device="cuda")
    ctx._check(ctx.lib.bevgen_op_gemm(ctx._h, _ptr(da), _ptr(dw), _ptr(db), _ptr(dr), _ptr(out), M, N, K, 1, mode, _stream()))
    o = out.cpu()
    err = float((o.double() - ref).norm() / ref.norm())
    assert err < 2e-6, (M, N, K, mode, err)
    print("digest", M, N, K, mode, hashlib.sha256(o.numpy().tobytes()).hexdigest())
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for flag in ("1", "0"):
        res = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, BEVGEN_GEMM_ROWSPLIT=flag), capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
        outs.append([l for l in res.stdout.splitlines() if l.startswith("digest")])
    assert len(outs[0]) == 3 and outs[0] == outs[1], (outs[0], outs[1])
