"""Diagnostic: per-segment cycle shares of the 256x256 GEMM (needs a -DLECCR_G256_STAMPS build in leccr_amd/libleccr_stamps.so)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from leccr_amd import _lib
_lib.LIB_PATH = os.environ.get('LECCR_LIB', os.path.join(os.path.dirname(_lib.LIB_PATH), 'libleccr_stamps.so'))
from leccr_amd import kernels as K
lib = _lib.lib()
lib.leccr_set_gemm_variant(2)
lib.leccr_debug_set_stamps.argtypes = [C.c_void_p]
names = ['L1 reads', 'bar', 'C00', 'bar', 'L2+C01+L3 (to L3 wait)', 'bar', 'C11+bar', 'L4 issue', 'vmcnt(8)', 'bar+C10+bar', 'epilogue', '-']
for name, tn, m, n, k in [('NT 4096^3', 0, 4096, 4096, 4096), ('NT qkv', 0, 147712, 2304, 768), ('NT fc2', 0, 147712, 768, 3072), ('TN dW fc1', 1, 3072, 768, 147712)]:
    st = torch.zeros(32 + 256 * 4, dtype=torch.int64, device='cuda')
    lib.leccr_debug_set_stamps(st.data_ptr())
    if tn:
        a = torch.randn(k, m, device='cuda').bfloat16(); b = torch.randn(k, n, device='cuda').bfloat16()
        out = torch.zeros(m, n, device='cuda'); f = lambda: K.gemm(a, b, trans_a=True, trans_b=True, out=out, accumulate=True)
    else:
        a = torch.randn(m, k, device='cuda').bfloat16(); b = torch.randn(n, k, device='cuda').bfloat16()
        out = torch.empty(m, n, device='cuda', dtype=torch.bfloat16); f = lambda: K.gemm(a, b, out=out)
    f(); f(); torch.cuda.synchronize()
    v = st.cpu().tolist()
    import collections
    blk = [(v[32+4*i+1]-v[32+4*i], v[32+4*i+2] & 15, v[32+4*i+3], v[32+4*i]) for i in range(256) if v[32+4*i+1]]
    t0 = min(b[3] for b in blk)
    byx = collections.defaultdict(list)
    for d, x, fi, st0 in blk: byx[x].append((d/100.0, (st0-t0)/100.0))
    print(name, 'per-XCC block duration us (min/mean/max), start offset max:', {x: (round(min(a for a,_ in l),1), round(sum(a for a,_ in l)/len(l),1), round(max(a for a,_ in l),1), round(max(b for _,b in l),1), len(l)) for x, l in sorted(byx.items())})
    for g in (0, 1):
        tot = sum(v[g*12:(g+1)*12])
        print(name, 'group', g, 'total cycles', tot, ' | '.join(f'{names[i]}: {100*v[g*12+i]/max(tot,1):.1f}%' for i in range(11)))
