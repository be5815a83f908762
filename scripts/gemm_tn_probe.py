import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from leccr_amd import kernels as K, _lib
_lib.lib().leccr_set_gemm_variant(2)
dev='cuda'
def run(name, tn, m, n, k, reps=5):
    if tn:
        a = torch.randn(k, m, device=dev).bfloat16(); b = torch.randn(k, n, device=dev).bfloat16()
        out = torch.zeros(m, n, device=dev); f = lambda: K.gemm(a, b, trans_a=True, trans_b=True, out=out, accumulate=True)
    else:
        a = torch.randn(m, k, device=dev).bfloat16(); b = torch.randn(n, k, device=dev).bfloat16()
        out = torch.empty(m, n, device=dev, dtype=torch.bfloat16); f = lambda: K.gemm(a, b, out=out)
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)/reps
    print(f'{name:28s} {ms*1e3:9.1f} us {2*m*n*k/ms/1e9:8.1f} TF/s', flush=True)
run('NT 4096^3', 0, 4096, 4096, 4096)
run('TN 4096^3 (256 tiles, 1 split)', 1, 4096, 4096, 4096)
run('TN 4096x4096 K=65536', 1, 4096, 4096, 65536)
run('TN 3072x768 K=147712', 1, 3072, 768, 147712)
run('TN 3072x768 K=16384*2', 1, 3072, 768, 32768)
run('TN 1024x1024 K=147712', 1, 1024, 1024, 147712)
run('TN 2048x2048 K=147712', 1, 2048, 2048, 147712)
