"""Per-item time of the fp8 4-wave NT kernel against K (M = 186880, N = 1024: 2920 tiles = 11.4 per CU): t(K) = rounds x (K-tiles x kt + E).
LECCR_LIB selects a diagnostic build (scripts/build_variant.sh).  Prints us per launch and per item for K = 512 .. 4096, and the fit."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from leccr_amd import _lib
if os.environ.get('LECCR_LIB'): _lib.LIB_PATH = os.environ['LECCR_LIB']
from leccr_amd import kernels as K
dev = 'cuda'
m, n = 186880, int(os.environ.get('N', 1024))
sites = K.Fp8Sites(dev, 8)
res = []
for k in (512, 1024, 2048, 4096):
    a = torch.randn(m, k, device=dev).bfloat16(); b = (torch.randn(n, k, device=dev) * 0.05).bfloat16()
    sa, sb = sites.new(0), sites.new(0)
    a8, b8 = sa.quantize(a), sb.quantize(b)
    out = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
    f = lambda: K.gemm(a8, b8, out=out, alpha_dev=(sa.inv_scale, sb.inv_scale))
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    tiles = (m // 256) * (n // 256)
    per_item = us / (tiles / 256.0)
    res.append((k, us, per_item))
    print(f'K={k:5d}  {us:8.1f} us  {2.0*m*n*k/us/1e6:7.1f} TF/s  per item {per_item:6.2f} us ({k // 128} K-tiles)', flush=True)
kt = (res[-1][2] - res[1][2]) / ((res[-1][0] - res[1][0]) / 128)
print(f'fit: K-tile {kt:.3f} us, per-item overhead {res[1][2] - (res[1][0] // 128) * kt:.2f} us')
