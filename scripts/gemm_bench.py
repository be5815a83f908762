"""GEMM micro-benchmark on MI355X: TFLOP/s of leccr_gemm on the step's main shapes (random data)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from leccr_amd import kernels as K

dev = 'cuda'
M = 256 * 577
SH = [('qkv', 0, M, 2304, 768), ('out', 0, M, 768, 768), ('fc1', 0, M, 3072, 768), ('fc2', 0, M, 768, 3072),
      ('dW qkv', 1, 2304, 768, M), ('dW out', 1, 768, 768, M), ('dW fc1', 1, 3072, 768, M), ('dW fc2', 1, 768, 3072, M),
      ('bert out', 0, 16384, 768, 768), ('bert qkv', 0, 16384, 2304, 768), ('bert fc1', 0, 16384, 3072, 768), ('bert fc2', 0, 16384, 768, 3072),
      ('bert dW fc1', 1, 3072, 768, 16384), ('bert dW out', 1, 768, 768, 16384), ('cap qkv', 0, 16384, 1536, 768), ('sq 4096', 0, 4096, 4096, 4096), ('sq 8192', 0, 8192, 8192, 8192)]
if os.environ.get('SMALL_M'):       # evaluation as yielded: 6 images x 577 tokens (ViT) and 6 x 64 caption / 32 x 32 text tokens (BERT) per forward
    m6 = int(os.environ['SMALL_M'])
    SH = [('qkv', 0, m6, 2304, 768), ('out', 0, m6, 768, 768), ('fc1', 0, m6, 3072, 768), ('fc2', 0, m6, 768, 3072),
          ('bert qkv', 0, 1024, 2304, 768), ('bert out', 0, 1024, 768, 768), ('bert fc1', 0, 1024, 3072, 768), ('bert fc2', 0, 1024, 768, 3072),
          ('cap qkv', 0, 384, 2304, 768), ('cap fc1', 0, 384, 3072, 768)]
reps = int(os.environ.get('REPS', 5))
from leccr_amd import _lib
if os.environ.get('LECCR_LIB'): _lib.LIB_PATH = os.environ['LECCR_LIB']
_lib.lib().leccr_set_gemm_variant(int(os.environ.get('VARIANT', 0)))
print('variant', os.environ.get('VARIANT', 0))
only = [x for x in os.environ.get('ONLY', '').split(',') if x]
for name, tn, m, n, k in SH:
    if only and name not in only: continue
    if tn:
        a = torch.randn(k, m, device=dev).bfloat16(); b = torch.randn(k, n, device=dev).bfloat16()
        out = torch.zeros(m, n, device=dev)
        f = lambda: K.gemm(a, b, trans_a=True, trans_b=True, out=out, accumulate=True)
    else:
        a = torch.randn(m, k, device=dev).bfloat16(); b = torch.randn(n, k, device=dev).bfloat16()
        out = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
        f = lambda: K.gemm(a, b, out=out)
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    lib_tf = float('nan')
    if os.environ.get('LIBCMP', '0') != '0':
        # measurement only: what the vendor library (torch.mm -> hipBLASLt / rocBLAS) reaches on the same operands, as a yardstick
        # for this kernel's headroom.  Not a product path.
        if tn:
            g = lambda: torch.mm(a.t(), b)            # [m,k]x[k,n], bf16 out (no f32 accumulate-into-C: an easier job)
        else:
            g = lambda: torch.mm(a, b.t())
        for _ in range(2): g()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps): g()
        e1.record(); torch.cuda.synchronize()
        lib_tf = 2 * m * n * k / (e0.elapsed_time(e1) / reps) / 1e9
    print(f'{name:12s} M={m:6d} N={n:5d} K={k:6d}  {ms*1e3:9.1f} us  {2*m*n*k/ms/1e9:8.1f} TFLOP/s   vendor lib {lib_tf:8.1f}', flush=True)
