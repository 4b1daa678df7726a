"""The Route-M self-attention of the bench step as an operator call (B = 16, H = 16, Nq = 1536, Nk_pad = 1568 incl. the null key, bias) - the launch tools/pmc_attn_sq.sh counts."""
import sys, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from bevgen_amd.runtime import Context
ctx = Context(None, precision="f16x3")
B, H, Nq, Nk = 16, 16, 1536, 1568
g = torch.Generator().manual_seed(0)
q = torch.nn.functional.normalize(torch.randn(B, H, Nq, 64, generator=g), dim=-1).cuda()
k = torch.nn.functional.normalize(torch.randn(B, H, Nk, 64, generator=g), dim=-1).cuda()
v = torch.randn(B, H, Nk, 64, generator=g).cuda()
bias = (torch.randn(Nq, Nk, generator=g) * 0.5).cuda()
bias[:, 1537:] = -1e30
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    out = ctx.op_attention(q, k, v, bias, 8.0)
ctx.synchronize()
print("ok", float(out.abs().mean()))
