"""LayerNorm forward/backward bandwidth at the step's shape ([256*577, 768] bf16)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from leccr_amd import kernels as K, _lib
if os.environ.get('LECCR_LIB'): _lib.LIB_PATH = os.environ['LECCR_LIB']
dev = 'cuda'
rows, d = 256 * 577, 768
x = torch.randn(rows, d, device=dev).bfloat16(); dy = torch.randn_like(x); dres = torch.randn_like(x)
g = torch.ones(d, device=dev); b = torch.zeros(d, device=dev); dg = torch.zeros(d, device=dev); db = torch.zeros(d, device=dev)
def t(f, reps=10):
    f(); f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
y, mean, rstd, _ = K.layernorm_fwd(x, g, b, 1e-5)
tf = t(lambda: K.layernorm_fwd(x, g, b, 1e-5))
tb = t(lambda: K.layernorm_bwd(dy, x, g, mean, rstd, dg, db, dres=dres))
cs = torch.zeros(d, device=dev)
tc = t(lambda: K.layernorm_bwd(dy, x, g, mean, rstd, dg, db, dres=dres, colsum=cs, colsum_of=1))
nb = rows * d * 2
print(f'LN fwd {tf*1e3:.1f} us  {2*nb/tf/1e9:.2f} TB/s (read x + write y) | LN bwd {tb*1e3:.1f} us  {4*nb/tb/1e9:.2f} TB/s (dy, x, dres -> dx) | with colsum(dres) {tc*1e3:.1f} us  {4*nb/tc/1e9:.2f} TB/s')
