"""Time the per-step weight-copy launches alone: leccr_transpose_cast_batched (bf16 path: [in,out] copies) and leccr_split3_weights_batched
(fp32 path: both split copies) over the Linear weights of ViT-B/16 and of mBERT's 12 layers (the shapes ParamArena.refresh hands them).
    python scripts/wcopy_bench.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from leccr_amd import kernels as K

DEV = 'cuda'
shapes = [(2304, 768), (768, 768), (3072, 768), (768, 3072)] * 12
ws = [torch.randn(s, device=DEV) for s in shapes]
n = sum(w.numel() for w in ws)


def table(cols, dsts):
    rows, tiles = [], 0
    for w, d in zip(ws, dsts):
        R, C = w.shape
        rows.append([tiles, w.data_ptr()] + [x.data_ptr() for x in d] + [R, C])
        tiles += ((R + 63) // 64) * ((C + 63) // 64)
    return torch.tensor(rows, dtype=torch.int64, device=DEV), tiles


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


lpt = [(torch.empty(w.shape[1], w.shape[0], dtype=torch.bfloat16, device=DEV),) for w in ws]
t, tiles = table(5, lpt)
us = timed(lambda: K.transpose_cast_batched(t, tiles))
print('transpose_cast_batched  %d weights %.1f M elements  %.1f us  %.2f TB/s (4 B read + 2 B written per element)' % (len(ws), n / 1e6, us, n * 6 / us / 1e6))
assert all(torch.equal(d[0], w.t().contiguous().to(torch.bfloat16)) for w, d in zip(ws[:4], lpt[:4]))
s3 = [(torch.empty(w.shape[0], 3 * w.shape[1], dtype=torch.bfloat16, device=DEV), torch.empty(w.shape[1], 3 * w.shape[0], dtype=torch.bfloat16, device=DEV)) for w in ws]
t3, tiles3 = table(6, s3)
us = timed(lambda: K.split3_weights_batched(t3, tiles3))
print('split3_weights_batched  %d weights %.1f M elements  %.1f us  %.2f TB/s (4 B read + 12 B written per element)' % (len(ws), n / 1e6, us, n * 16 / us / 1e6))
for w, (g, tt) in zip(ws[:4], s3[:4]):
    assert torch.equal(g, K.split3(w, K.FORM_G)) and torch.equal(tt, K.split3(w, K.FORM_X, transpose=True))
print('bit-exact vs the per-tensor passes')
