"""Diagnostic: where one block of the 256x256 GEMM spends its time - K loop / epilogue / refill, per XCD, and the in-kernel clock
(needs a -DLECCR_G256_STAMPS build: scripts/build_variant.sh stamps -DLECCR_G256_STAMPS gemm256)."""
import ctypes as C, os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from leccr_amd import _lib
_lib.LIB_PATH = os.environ.get('LECCR_LIB', 'gpurun_variants/libleccr_stamps.so')
from leccr_amd import kernels as K
lib = _lib.lib()
lib.leccr_set_gemm_variant(2)
lib.leccr_debug_set_stamps.argtypes = [C.c_void_p]
M = 256 * 577
for name, tn, m, n, k in [('NT 4096^3', 0, 4096, 4096, 4096), ('NT 8192^3', 0, 8192, 8192, 8192), ('NT qkv', 0, M, 2304, 768), ('NT out', 0, M, 768, 768),
                          ('NT fc1', 0, M, 3072, 768), ('NT fc2', 0, M, 768, 3072), ('TN dW fc1', 1, 3072, 768, M)]:
    st = torch.zeros(32 + 256 * 8, dtype=torch.int64, device='cuda')
    if tn:
        a = torch.randn(k, m, device='cuda').bfloat16(); b = torch.randn(k, n, device='cuda').bfloat16()
        out = torch.zeros(m, n, device='cuda'); f = lambda: K.gemm(a, b, trans_a=True, trans_b=True, out=out, accumulate=True)
    else:
        a = torch.randn(m, k, device='cuda').bfloat16(); b = torch.randn(n, k, device='cuda').bfloat16()
        out = torch.empty(m, n, device='cuda', dtype=torch.bfloat16); f = lambda: K.gemm(a, b, out=out)
    lib.leccr_debug_set_stamps(None)
    for _ in range(20): f()                       # reach the sustained clock
    lib.leccr_debug_set_stamps(st.data_ptr())
    f(); torch.cuda.synchronize()
    v = st.cpu().tolist()
    rows = [v[32 + 8 * i: 40 + 8 * i] for i in range(256) if v[32 + 8 * i + 1]]
    t0, t1 = min(r[0] for r in rows), max(r[1] for r in rows)
    nkt = (k + 63) // 64
    print(f'{name}: kernel span {(t1 - t0) / 100:.1f} us, {len(rows)} blocks, items/block {min(r[5] for r in rows)}-{max(r[5] for r in rows)}, K-tiles/item {nkt if not tn else "split"}')
    byx = collections.defaultdict(list)
    for r in rows: byx[r[6]].append(r)
    for x, l in sorted(byx.items()):
        n_ = len(l)
        dur = [(r[1] - r[0]) / 100 for r in l]
        items = sum(r[5] for r in l)
        print(f'   xcc {x}: blocks {n_:3d} dur us min/mean/max {min(dur):7.1f} {sum(dur)/n_:7.1f} {max(dur):7.1f}  start spread {(max(r[0] for r in l) - t0) / 100:5.1f}'
              f'  per item us: K loop {sum(r[2] for r in l) / items / 100:6.2f} epilogue {sum(r[3] for r in l) / items / 100:6.2f} refill {sum(r[4] for r in l) / items / 100:5.2f}'
              f'  clock {sum(r[7] / (r[1] - r[0]) / 10 for r in l) / n_:.2f} GHz')
