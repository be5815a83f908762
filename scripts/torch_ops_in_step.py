"""Which torch (aten) device ops still run inside one train step, and from where.
    python scripts/torch_ops_in_step.py [--batch 32]
Runs two warm-up steps, then one step under a TorchDispatchMode that records every aten op touching a CUDA tensor with the innermost
leccr_amd / bench.py frame that issued it (views and metadata ops are left out)."""
import argparse, collections, os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode

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

SKIP = ('view', 'reshape', 'expand', 'permute', 'transpose', 'slice', 'select', 'as_strided', 'alias', 'detach', 'unsqueeze', 'squeeze',
        't.default', 'empty', 'size', 'stride', 'is_', 'unbind', 'split', 'record_stream', '_unsafe_view', 'lift_fresh', 'numel', 'sym_')
rows = collections.Counter()


class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        flat = [x for x in list(args) + list((kwargs or {}).values()) if isinstance(x, torch.Tensor)]
        if flat and any(t.is_cuda for t in flat) and not any(s in name for s in SKIP):
            fr = [f for f in traceback.extract_stack() if ('leccr_amd' in f.filename or f.filename.endswith('bench.py')) and 'scripts' not in f.filename]
            where = f'{os.path.relpath(fr[-1].filename)}:{fr[-1].lineno} {fr[-1].name}' if fr else '(autograd engine / torch internals)'
            rows[(name, str([tuple(t.shape) for t in flat])[:70], where)] += 1
        return func(*args, **(kwargs or {}))


with Log():
    bench.train_step(model, batch, fwd, opt, sched)
    torch.cuda.synchronize()
for (name, shapes, where), n in sorted(rows.items(), key=lambda kv: -kv[1]):
    print(f'{n:4d}  {name:34s} {shapes:70s} {where}')
print('total aten device ops in the step:', sum(rows.values()))
