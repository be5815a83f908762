"""Attention micro-benchmark (bf16, head 64): forward / backward TFLOP/s at the step's shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from leccr_amd import kernels as K, _lib
if os.environ.get('LECCR_LIB'): _lib.LIB_PATH = os.environ['LECCR_LIB']
dev = 'cuda'
def t(f, reps=5):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
SH = [('warm', 256, 12, 576)] + [('vit %d' % L, 256, 12, L) for L in (512, 560, 576, 577, 578, 592, 608, 640)] if os.environ.get('SWEEP') else [('warm', 256, 12, 576), ('vit 577', 256, 12, 577), ('vit 576', 256, 12, 576), ('bert 32', 512, 12, 32), ('bert 64', 256, 12, 64)]
for name, b, h, L in SH:
    d = h * 64
    qkv = torch.randn(b, L, 3 * d, device=dev)
    if os.environ.get("DT", "bf16") == "bf16": qkv = qkv.bfloat16()
    q, k, v = qkv[..., :d], qkv[..., d:2*d], qkv[..., 2*d:]
    o, lse = K.attn_fwd(q, k, v, h)
    do = torch.randn_like(o)
    dqkv = torch.empty_like(qkv)
    fl = 4.0 * b * h * L * L * 64
    tf = t(lambda: K.attn_fwd(q, k, v, h))
    tb = t(lambda: K.attn_bwd(q, k, v, o, do, lse, h, dq=dqkv[..., :d], dk=dqkv[..., d:2*d], dv=dqkv[..., 2*d:]))
    print(f'{name:10s} fwd {tf*1e3:8.1f} us {fl/tf/1e9:7.1f} TF/s | bwd {tb*1e3:8.1f} us {2.5*fl/tb/1e9:7.1f} TF/s', flush=True)
