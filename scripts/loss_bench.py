"""Loss kernels at the gathered batch sizes of 1..8 ranks (G = 256 * W): forward + backward time of each loss term."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from leccr_amd import ops
dev = 'cuda'
def t(f, reps=3):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
n, D, vw = 4, 256, 256
for W in (1, 2, 4, 8):
    G = 256 * W
    nrm = lambda x: torch.nn.functional.normalize(x, dim=-1)
    a = nrm(torch.randn(G, D, device=dev)).requires_grad_(True)
    b = nrm(torch.randn(G, D, device=dev)).requires_grad_(True)
    idx = torch.arange(G, device=dev)
    temp = torch.tensor(0.07, device=dev, requires_grad=True)
    cq = nrm(torch.randn(G, n, D, device=dev)).requires_grad_(True)
    cq_loc = nrm(torch.randn(256, n, D, device=dev)).requires_grad_(True)
    tl = nrm(torch.randn(256, D, device=dev)).requires_grad_(True)
    pc = torch.randn(G, vw, device=dev, requires_grad=True)
    pv = torch.randn(G, vw, device=dev, requires_grad=True)
    lr = None if W == 1 else (0, 256)          # this rank's rows of the gathered batch
    def itc(): ops.SimInfoNCEFn.apply(a, b, temp, idx, True, 0.07, lr).backward()
    def cv(): ops.SimInfoNCEFn.apply(pc, pv, None, idx, False, 1.0, lr).backward()
    def cap(): ops.CapSimFn.apply(cq_loc, tl, temp, 0.07).backward()
    def dstl(): ops.DstlFn.apply(a, cq, a, b, 0.4, lr).backward()
    print(f'W={W} G={G}: itc {t(itc):7.3f} ms (x3/step) | cv {t(cv):7.3f} | capsim(local) {t(cap):7.3f} (x2) | dstl {t(dstl):7.3f}', flush=True)
