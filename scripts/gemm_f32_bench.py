"""fp32 products of the step's shapes on the bf16 matrix cores (hi / lo operand pairs, kernels.gemm) against the scalar f32 kernel:
split passes, the bf16 launch over 3 K, and what the pair of them delivers in fp32-product TFLOP/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from leccr_amd import kernels as K
dev = 'cuda'
def t(f, reps=5):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
M = 256 * 577
SH = [('qkv', M, 2304, 768, False), ('out', M, 768, 768, False), ('fc1', M, 3072, 768, False), ('fc2', M, 768, 3072, False),
      ('dW qkv', 2304, 768, M, True), ('dW out', 768, 768, M, True), ('dW fc1', 3072, 768, M, True), ('dW fc2', 768, 3072, M, True),
      ('bert fc1', 16384, 3072, 768, False)]
print(f'{"shape":10s} {"M":>7s} {"N":>5s} {"K":>7s} | split A us | split B us | bf16 launch over 3K us (issued TF/s) | fp32-product TF/s: launch only / with both splits | scalar f32 kernel TF/s')
for name, m, n, k, tn in SH:
    if tn:
        a, b = torch.randn(k, m, device=dev), torch.randn(k, n, device=dev)
        ta = t(lambda: K.split3(a, K.FORM_G)); tb = t(lambda: K.split3(b, K.FORM_X))
        a3, b3 = K.split3(a, K.FORM_G), K.split3(b, K.FORM_X)
        out = torch.zeros(m, n, device=dev)
        tg = t(lambda: K.gemm(a, b, trans_a=True, trans_b=True, out=out, accumulate=True, a3=(a3, K.FORM_G), b3=(b3, K.FORM_X)))
        prev = K.set_f32_mfma(False)
        ts = t(lambda: K.gemm(a[:k // 16], b[:k // 16], trans_a=True, trans_b=True, out=out, accumulate=True), reps=2) * 16
        K.set_f32_mfma(prev)
    else:
        a, b = torch.randn(m, k, device=dev), torch.randn(n, k, device=dev) * 0.05
        ta = t(lambda: K.split3(a, K.FORM_X)); tb = t(lambda: K.split3(b, K.FORM_G))
        a3, b3 = K.split3(a, K.FORM_X), K.split3(b, K.FORM_G)
        tg = t(lambda: K.gemm(a, b, a3=(a3, K.FORM_X), b3=(b3, K.FORM_G)))
        prev = K.set_f32_mfma(False)
        ts = t(lambda: K.gemm(a[:m // 16], b), reps=2) * 16
        K.set_f32_mfma(prev)
    fl = 2.0 * m * n * k
    print(f'{name:10s} {m:7d} {n:5d} {k:7d} | {ta*1e3:9.1f} | {tb*1e3:9.1f} | {tg*1e3:9.1f} ({3*fl/tg/1e9:7.1f}) | {fl/tg/1e9:7.1f} / {fl/(tg+ta+tb)/1e9:7.1f} | {fl/ts/1e9:6.1f}', flush=True)
