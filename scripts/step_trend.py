"""Per-step wall time of the bench workload from a cold start (is the warm-up long enough?)."""
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
model, cfg = bench.build_model(args, dev)
opt = attach_grad_sync(create_optimizer(cfg['optimizer'], model))
sched = create_scheduler(dict(cfg['schedular'], step_per_epoch=1000), opt)
cpu = make_batch(args.batch)
batch = {'image': cpu['image'].to(dev), 'text': [t.to(dev) for t in cpu['text']], 'captions': cpu['captions'].to(dev), 'idx': cpu['idx'].to(dev)}
ts = []
for i in range(16):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    bench.train_step(model, batch, opt, sched)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print(' '.join(f'{t:.1f}' for t in ts))
