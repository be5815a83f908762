"""fp8 vs bf16 GEMM on the ViT-L/14 + XLM-R-large shapes of BASELINE configs[4] (b = 256: M = 256 x 730 tokens) and on the
ViT-B/16 shapes, random data: TFLOP/s of leccr_gemm with LECCR_FP8 operands (4-wave 256x256 kernel, scaled MFMA) against the
bf16 production kernel, and the cost of the quantisation passes that feed it."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from leccr_amd import kernels as K

dev = 'cuda'
reps = int(os.environ.get('REPS', 10))
SH = [('L14 qkv', 186880, 3072, 1024), ('L14 out', 186880, 1024, 1024), ('L14 fc1', 186880, 4096, 1024), ('L14 fc2', 186880, 1024, 4096),
      ('B16 qkv', 147712, 2304, 768), ('B16 fc1', 147712, 3072, 768), ('B16 fc2', 147712, 768, 3072), ('sq 8192', 8192, 8192, 8192)]


def timeit(f):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


sites = K.Fp8Sites(dev, 64)
for name, m, n, k in SH:
    a = torch.randn(m, k, device=dev).bfloat16(); b = (torch.randn(n, k, device=dev) * 0.05).bfloat16()
    out = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
    sa, sb = sites.new(0), sites.new(0)
    a8, b8 = sa.quantize(a), sb.quantize(b)
    t_bf = timeit(lambda: K.gemm(a, b, out=out))
    t_f8 = timeit(lambda: K.gemm(a8, b8, out=out, alpha_dev=(sa.inv_scale, sb.inv_scale)))
    t_q = timeit(lambda: sa.quantize(a, out=a8))
    fl = 2.0 * m * n * k
    print(f'{name:9s} M={m:6d} N={n:5d} K={k:5d}  bf16 {t_bf*1e3:8.1f} us {fl/t_bf/1e9:7.1f} TF/s | fp8 {t_f8*1e3:8.1f} us {fl/t_f8/1e9:7.1f} TF/s '
          f'(x{t_bf/t_f8:4.2f}) | quantise A {t_q*1e3:7.1f} us = {m*k*3/t_q/1e9:5.2f} TB/s', flush=True)
