"""Run the ViT attention shape a few times (for rocprofv3 counter collection).  usage: attn_one.py reps [L]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from leccr_amd import kernels as K
reps = int(sys.argv[1]); L = int(sys.argv[2]) if len(sys.argv) > 2 else 577
b, h, d = 256, 12, 768
qkv = torch.randn(b, L, 3 * d, device='cuda')
if os.environ.get('DT', 'bf16') == 'bf16': qkv = qkv.bfloat16()
q, k, v = qkv[..., :d], qkv[..., d:2*d], qkv[..., 2*d:]
o, lse = K.attn_fwd(q, k, v, h)
do = torch.randn_like(o)
dqkv = torch.empty_like(qkv)
for _ in range(reps):
    o, lse = K.attn_fwd(q, k, v, h)
    K.attn_bwd(q, k, v, o, do, lse, h, dq=dqkv[..., :d], dk=dqkv[..., d:2*d], dv=dqkv[..., 2*d:])
torch.cuda.synchronize()
