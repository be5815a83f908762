"""Run one GEMM shape a few times (for rocprofv3 counter collection).  usage: gemm_one.py tn M N K variant reps"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from leccr_amd import kernels as K, _lib
tn, m, n, k, variant, reps = [int(x) for x in sys.argv[1:7]]
_lib.lib().leccr_set_gemm_variant(variant)
dev = 'cuda'
if tn:
    a = torch.randn(k, m, device=dev).bfloat16(); b = torch.randn(k, n, device=dev).bfloat16()
    out = torch.zeros(m, n, device=dev)
    f = lambda: K.gemm(a, b, trans_a=True, trans_b=True, out=out, accumulate=True)
else:
    a = torch.randn(m, k, device=dev).bfloat16(); b = torch.randn(n, k, device=dev).bfloat16()
    out = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
    f = lambda: K.gemm(a, b, out=out)
for _ in range(reps): f()
torch.cuda.synchronize()
