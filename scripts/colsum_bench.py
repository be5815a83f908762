"""Column-sum (bias gradient) bandwidth at the step's shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from leccr_amd import kernels as K, _lib
if os.environ.get('LECCR_LIB'): _lib.LIB_PATH = os.environ['LECCR_LIB']
for M, N in [(256 * 577, 3072), (256 * 577, 2304), (16384, 3072)]:
    x = torch.randn(M, N, device='cuda').bfloat16(); out = torch.zeros(N, device='cuda')
    for _ in range(3): K.colsum(x, out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): K.colsum(x, out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    ref = x.float().sum(0) * 13
    err = float((out - ref).abs().max() / ref.abs().max())
    print(f'colsum [{M}, {N}] bf16: {ms*1e3:.1f} us  {M*N*2/ms/1e9:.2f} TB/s  rel err {err:.1e}')
