"""Per-step wall time and allocator state of the bench workload from a cold start (is the warm-up long enough?  does memory
grow with the step count - side streams, record_stream, the gather cache?)."""
import sys, time, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
sys.argv = ['bench.py']
args = bench.parse()
from leccr_amd.optim import create_optimizer
from leccr_amd.parallel import attach_grad_sync
from leccr_amd.scheduler import create_scheduler
from leccr_amd.synthetic import make_batch
dev = torch.device('cuda', 0)
model, cfg, batch, fwd, _ = bench.build_workload(args, dev, 0)
opt = attach_grad_sync(create_optimizer(cfg['optimizer'], model)).enable_backward_fusion()
sched = create_scheduler(dict(cfg['schedular'], step_per_epoch=1000), opt)
ts = []
n = int(os.environ.get('STEPS', 48))
for i in range(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    bench.train_step(model, batch, fwd, opt, sched)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    if i % 8 == 7:
        print(f'step {i + 1:3d}: last 8 steps ' + ' '.join(f'{t:.1f}' for t in ts[-8:]) +
              f' ms | allocated {torch.cuda.memory_allocated() / 2**30:.2f} GiB reserved {torch.cuda.memory_reserved() / 2**30:.2f} GiB', flush=True)
