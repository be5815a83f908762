"""Diagnostic: where a wave of the flash forward spends a key tile (needs a -DLECCR_ATTN_STAMPS build:
scripts/build_variant.sh attn_stamps "-DLECCR_ATTN_STAMPS -fno-slp-vectorize" attention_mfma)."""
import ctypes as C, os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from leccr_amd import _lib
_lib.LIB_PATH = os.environ.get('LECCR_LIB', 'gpurun_variants/libleccr_attn_stamps.so')
from leccr_amd import kernels as K
lib = _lib.lib()
lib.leccr_debug_set_attn_stamps.argtypes = [C.c_void_p]
b, h, L = 256, 12, int(os.environ.get('L', 577))
d = h * 64
qkv = torch.randn(b, L, 3 * d, device='cuda').bfloat16()
q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
NW, NB = 8, 4096
st = torch.zeros(NB * NW * 16 * 8, dtype=torch.int64, device='cuda')
lib.leccr_debug_set_attn_stamps(None)
for _ in range(10): K.attn_fwd(q, k, v, h)
torch.cuda.synchronize()
lib.leccr_debug_set_attn_stamps(st.data_ptr())
K.attn_fwd(q, k, v, h); torch.cuda.synchronize()
lib.leccr_debug_set_attn_stamps(None)
a = st.cpu().view(NB, NW, 16, 8).numpy()
names = ['top -> past barrier (DMA wait + barrier)', 'barrier -> K fragments ready (stage issue + LDS read)', 'K ready -> V ready (S MFMA, softmax, V read wait)',
         'V ready -> PV MFMAs issued', 'tile end -> next tile top']
import numpy as np
nt = (L + 63) // 64
full = a[:, :, 1:nt - 2, :]                      # steady-state tiles
ok = full[..., 4] > 0
seg = [full[..., i + 1] - full[..., i] for i in range(4)]
nxt = a[:, :, 2:nt - 1, 0] - a[:, :, 1:nt - 2, 4]
tot = a[:, :, 2:nt - 1, 0] - a[:, :, 1:nt - 2, 0]
ok2 = ok & (a[:, :, 2:nt - 1, 4] > 0)
print(f'flash_fwd b{b} h{h} L{L}: s_memtime ticks (100 MHz) per steady-state tile, mean over {int(ok2.sum())} (wave, tile) samples')
for n_, s_ in zip(names, seg + [nxt]):
    print(f'  {n_:58s} {s_[ok2].mean():8.2f} ticks  ({100 * s_[ok2].mean() / tot[ok2].mean():5.1f} %)   p10 {np.percentile(s_[ok2], 10):6.1f} p90 {np.percentile(s_[ok2], 90):6.1f}')
print(f'  tile period {tot[ok2].mean():8.2f} ticks = {tot[ok2].mean() * 10:.0f} ns')
# waves per SIMD alive together: group the samples of tile 3 by (xcc, hw_id simd/cu/se)
hw = a[:, :, 3, 5]; xc = a[:, :, 3, 6]
key = (xc << 32) | (hw & 0xFFF0)
t0 = a[:, :, 3, 0]; t1 = a[:, :, 3, 4]
by = collections.defaultdict(list)
for kk, s0, s1 in zip(key.ravel(), t0.ravel(), t1.ravel()):
    if s1 > 0: by[int(kk)].append((int(s0), int(s1)))
ov = []
for kk, l in by.items():
    for (s0, s1) in l:
        ov.append(sum(1 for (u0, u1) in l if u0 < s1 and u1 > s0))
print(f'  waves on the same SIMD whose tile 3 overlaps a wave\'s tile 3 (incl. itself): mean {np.mean(ov):.2f}; SIMDs seen {len(by)}')
