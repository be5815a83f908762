"""Which torch device ops (copies, fills, elementwise adds) still run inside one train step, and from where.
    python scripts/torch_ops_in_step.py [--batch 32]
Runs two warm-up steps and one step under torch.profiler (with Python stacks), then prints every at::native / copy kernel's op,
count, input shapes and the innermost leccr_amd / bench frame that issued it."""
import argparse, collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=32)
a = ap.parse_args()
import bench
args = argparse.Namespace(model='image', clip='ViT-B/16', text='mbert', dtype='bf16', batch=a.batch)
dev = torch.device('cuda', 0)
model, cfg, batch, fwd, what = bench.build_workload(args, dev, 0)
from leccr_amd.optim import create_optimizer
from leccr_amd.scheduler import create_scheduler
opt = create_optimizer(cfg['optimizer'], model).enable_backward_fusion()
sched = create_scheduler(dict(cfg['schedular'], step_per_epoch=1000), opt)
for _ in range(2):
    bench.train_step(model, batch, fwd, opt, sched)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    bench.train_step(model, batch, fwd, opt, sched)
    torch.cuda.synchronize()
rows = collections.Counter()
for e in prof.events():
    if e.device_type.name != 'CPU' or not e.name.startswith('aten::'):
        continue
    if not e.kernels:
        continue
    frame = next((s for s in e.stack if ('leccr_amd' in s or 'bench.py' in s) and 'torch/' not in s), e.stack[0] if e.stack else '?')
    rows[(e.name, str(e.input_shapes)[:80], frame.strip()[-110:])] += 1
for (name, shapes, frame), n in sorted(rows.items(), key=lambda kv: -kv[1]):
    print(f'{n:4d}  {name:28s} {shapes:80s} {frame}')
print('kernels by name:')
kc = collections.Counter()
for e in prof.events():
    if e.device_type.name == 'CUDA':
        kc[e.name[:90]] += 1
for k, n in kc.most_common(40):
    print(f'{n:5d}  {k}')
