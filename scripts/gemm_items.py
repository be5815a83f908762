"""Per-item time of the bf16 8-wave NT kernel against K (M = 147712, N = 768: 1731 tiles, 7 rounds on 256 CUs): t(K) = rounds x (K-tiles x kt + E).
LECCR_LIB selects a diagnostic build (scripts/build_variant.sh g256nostore "-DG256_EXP_NOSTORE" gemm256)."""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from leccr_amd import _lib
if os.environ.get('LECCR_LIB'): _lib.LIB_PATH = os.environ['LECCR_LIB']
from leccr_amd import kernels as K
dev = 'cuda'
m, n = int(os.environ.get('M', 147712)), int(os.environ.get('N', 768))
res = []
for k in (384, 768, 1536, 3072):
    a = torch.randn(m, k, device=dev).bfloat16(); b = (torch.randn(n, k, device=dev) * 0.05).bfloat16()
    out = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
    f = lambda: K.gemm(a, b, out=out)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    tiles = math.ceil(m / 256) * math.ceil(n / 256)
    rounds = tiles / 256.0
    res.append((k, us, us / rounds))
    print(f'K={k:5d}  {us:8.1f} us  {2.0*m*n*k/us/1e6:7.1f} TF/s  per item {us / rounds:6.2f} us ({k // 64} K-tiles; kernel {_lib.lib().leccr_last_gemm_kernel()})', flush=True)
kt = (res[-1][2] - res[1][2]) / ((res[-1][0] - res[1][0]) / 64)
print(f'fit: K-tile {kt:.3f} us, per-item overhead {res[1][2] - (res[1][0] // 64) * kt:.2f} us')
