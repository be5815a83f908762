"""Weight-gradient products at the config-E geometry (ViT-L/14@224 b 256: 65792 token rows, d 1024): bf16 4-wave TN kernel against the
fp8 form of the same kernel (e5m2 dY, e4m3 X).  TFLOP/s with random data on one MI355X."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from leccr_amd import kernels as K, _lib

dev = 'cuda'
ROWS = 256 * 257
SH = [('dW qkv', 3072, 1024, ROWS), ('dW out', 1024, 1024, ROWS), ('dW fc1', 4096, 1024, ROWS), ('dW fc2', 1024, 4096, ROWS),
      ('E dW qkv', 3072, 1024, 186880), ('E dW out', 1024, 1024, 186880), ('E dW fc1', 4096, 1024, 186880), ('E dW fc2', 1024, 4096, 186880),
      ('xlmr dW fc1', 4096, 1024, 16384), ('xlmr dW out', 1024, 1024, 16384), ('B16 dW fc1', 3072, 768, 256 * 577)]
reps = int(os.environ.get('REPS', 5))
sites = K.Fp8Sites(dev, 4)
sg, sx = sites.new(1), sites.new(0)
for name, m, n, k in SH:
    dy = (torch.randn(k, m, device=dev) * 1e-3).bfloat16()
    x = torch.randn(k, n, device=dev).bfloat16()
    g8, x8 = sg.quantize(dy), sx.quantize(x)
    out = torch.zeros(m, n, device=dev)
    res = []
    bsum = torch.zeros(m, device=dev)
    for f in (lambda: K.gemm(dy, x, trans_a=True, trans_b=True, out=out, accumulate=True),
              lambda: K.gemm(g8, x8, trans_a=True, trans_b=True, out=out, accumulate=True, alpha_dev=(sg.inv_scale, sx.inv_scale)),
              lambda: K.gemm(g8, x8, trans_a=True, trans_b=True, out=out, accumulate=True, alpha_dev=(sg.inv_scale, sx.inv_scale), asum=bsum),
              lambda: K.colsum(dy, bsum)):
        for _ in range(2): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): f()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        res.append((ms, _lib.lib().leccr_last_gemm_kernel()))
    fl = 2 * m * n * k
    print(f'{name:12s} M={m:5d} N={n:5d} K={k:6d}  bf16 (kernel {res[0][1]}) {res[0][0]*1e3:8.1f} us {fl/res[0][0]/1e9:7.1f} TF/s   '
          f'fp8 (kernel {res[1][1]}) {res[1][0]*1e3:8.1f} us {fl/res[1][0]/1e9:7.1f} TF/s   x{res[0][0]/res[1][0]:.2f}   '
          f'fp8 + bias sums {res[2][0]*1e3:8.1f} us   (bf16 colsum pass alone {res[3][0]*1e3:6.1f} us)', flush=True)
