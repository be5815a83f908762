"""Forward / backward error of the bf16 flash kernels against an f64 reference on the same bf16 inputs (A/B of library builds:
LECCR_LIB=...).  SCALE sets the spread of the raw scores (1: near-uniform rows, 4: peaked rows)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from leccr_amd import kernels as K, _lib
if os.environ.get('LECCR_LIB'): _lib.LIB_PATH = os.environ['LECCR_LIB']
torch.manual_seed(0)
b, h, L = 4, 12, 577
d = h * 64
for scale in (1.0, 2.0, 4.0):
    qkv = (torch.randn(b, L, 3 * d, device='cuda') * torch.tensor([scale] * d + [scale] * d + [1.0] * d, device='cuda')).bfloat16()
    q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
    o, lse = K.attn_fwd(q, k, v, h)
    do = torch.randn_like(o)
    dqkv = torch.empty_like(qkv)
    K.attn_bwd(q, k, v, o, do, lse, h, dq=dqkv[..., :d], dk=dqkv[..., d:2 * d], dv=dqkv[..., 2 * d:])
    qr, kr, vr = (t.double().view(b, L, h, 64).transpose(1, 2).detach().requires_grad_(True) for t in (q, k, v))
    s = qr @ kr.transpose(-1, -2) / 8.0
    ref = (torch.softmax(s, -1) @ vr)
    (ref * do.double().view(b, L, h, 64).transpose(1, 2)).sum().backward()
    refo = ref.transpose(1, 2).reshape(b, L, d)
    def rel(a, r): return float((a.double() - r).norm() / r.norm())
    g = [t.grad.transpose(1, 2).reshape(b, L, d) for t in (qr, kr, vr)]
    print(f'scale {scale}: rel L2 err  o {rel(o, refo):.3e}  lse {float((lse.double() - torch.logsumexp(s, -1).view(b, h, L)).abs().max()):.3e}  '
          f'dq {rel(dqkv[..., :d], g[0]):.3e} dk {rel(dqkv[..., d:2*d], g[1]):.3e} dv {rel(dqkv[..., 2*d:], g[2]):.3e}', flush=True)
