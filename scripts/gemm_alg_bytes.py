"""Algorithmic HBM bytes of the persistent-GEMM launches of one train step (operands read once, outputs written once; the
accumulating f32 weight-gradient output counted read + written), to hold against profiles/r01_gemm256_hbm_traffic.json."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from leccr_amd import kernels as K, _lib
sys.argv = ['bench.py', '--no-cpu-baseline']
args = bench.parse()
model, cfg, batch, fwd, _ = bench.build_workload(args, torch.device('cuda', 0), 0)
from leccr_amd.optim import create_optimizer
from leccr_amd.scheduler import create_scheduler
from leccr_amd.synthetic import make_batch
opt = create_optimizer(cfg['optimizer'], model)
sched = create_scheduler(dict(cfg['schedular'], step_per_epoch=1000), opt); sched.step()
bench.train_step(model, batch, fwd, opt, sched)
tot = {'bytes': 0, 'n': 0, 'flop': 0}
orig = K.gemm
def counted(a, b, **kw):
    out = orig(a, b, **kw)
    if _lib.lib().leccr_last_gemm_kernel() == 2:
        ta, tb = kw.get('trans_a', False), kw.get('trans_b', False)
        M, Kd = (a.shape[1], a.shape[0]) if ta else a.shape
        N = b.shape[1] if tb else b.shape[0]
        by = (M * Kd + N * Kd) * a.element_size() + M * N * out.element_size() * (2 if kw.get('accumulate') else 1)
        for k in ('aux', 'aux_out', 'residual'):
            if kw.get(k) is not None: by += M * N * kw[k].element_size()
        tot['bytes'] += by; tot['n'] += 1; tot['flop'] += 2 * M * N * Kd
    return out
K.gemm = counted
from leccr_amd import ops
ops.K.gemm = counted
bench.train_step(model, batch, fwd, opt, sched)
torch.cuda.synchronize()
print(f"persistent-GEMM launches per step {tot['n']}, algorithmic bytes per launch {tot['bytes']/tot['n']/1e6:.1f} MB, GFLOP per launch {tot['flop']/tot['n']/1e9:.1f}")
