"""Training-script utilities with the reference's names (utils/__init__.py): SmoothedValue, MetricLogger,
distributed helpers.  Host-side bookkeeping only."""
import datetime
import os
import time
from collections import defaultdict, deque

import torch
import torch.distributed as dist


class SmoothedValue:
    """utils/__init__.py:14-74: window median/avg + global average."""

    def __init__(self, window_size=20, fmt=None):
        self.deque = deque(maxlen=window_size)
        self.total, self.count = 0.0, 0
        self.fmt = fmt or '{median:.4f} ({global_avg:.4f})'

    def update(self, value, n=1):
        self.deque.append(value)
        self.count += n
        self.total += value * n

    def synchronize_between_processes(self):
        if not is_dist_avail_and_initialized():
            return
        t = torch.tensor([self.count, self.total], dtype=torch.float64,
                         device='cuda' if dist.get_backend() == 'nccl' else 'cpu')
        dist.barrier()
        dist.all_reduce(t)
        self.count, self.total = int(t[0]), float(t[1])

    @property
    def median(self):
        return torch.tensor(list(self.deque)).median().item()

    @property
    def avg(self):
        return torch.tensor(list(self.deque), dtype=torch.float32).mean().item()

    @property
    def global_avg(self):
        return self.total / max(self.count, 1)

    @property
    def value(self):
        return self.deque[-1]

    def __str__(self):
        if not self.deque:          # nothing logged yet (the train loop reports losses one step late, see pipeline.py)
            return '-'
        return self.fmt.format(median=self.median, avg=self.avg, global_avg=self.global_avg, value=self.value)


class MetricLogger:
    """utils/__init__.py:77-229 (the parts train() uses)."""

    def __init__(self, delimiter='\t'):
        self.meters = defaultdict(SmoothedValue)
        self.delimiter = delimiter

    def add_meter(self, name, meter):
        self.meters[name] = meter

    def update(self, **kwargs):
        for k, v in kwargs.items():
            self.meters[k].update(float(v))

    def global_avg(self):
        return self.delimiter.join('{}: {:.4f}'.format(n, m.global_avg) for n, m in self.meters.items())

    def synchronize_between_processes(self):
        for m in self.meters.values():
            m.synchronize_between_processes()

    def log_every(self, iterable, print_freq, header=''):
        start = time.time()
        n = len(iterable) if hasattr(iterable, '__len__') else -1
        for i, obj in enumerate(iterable):
            yield obj
            if i % print_freq == 0 or i == n - 1:
                print(self.delimiter.join([header, f'[{i}/{n}]'] + [f'{k}: {m}' for k, m in self.meters.items()] +
                                          [f'elapsed: {datetime.timedelta(seconds=int(time.time() - start))}']))


def is_dist_avail_and_initialized():
    return dist.is_available() and dist.is_initialized()


def get_world_size():
    return dist.get_world_size() if is_dist_avail_and_initialized() else 1


def get_rank():
    return dist.get_rank() if is_dist_avail_and_initialized() else 0


def is_main_process():
    return get_rank() == 0


def init_distributed_mode(args):
    """utils/__init__.py:306-328: env:// rendezvous, backend 'nccl' (= RCCL on ROCm), one process per GPU."""
    if 'RANK' in os.environ and 'WORLD_SIZE' in os.environ:
        args.rank, args.world_size = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
        args.gpu = int(os.environ['LOCAL_RANK'])
    else:
        print('Not using distributed mode')
        args.distributed = False
        return
    args.distributed = True
    from .parallel import setup_rccl_env
    setup_rccl_env()                       # the same RCCL environment as leccr_amd.parallel.init_distributed_mode
    torch.cuda.set_device(args.gpu)
    from . import ops
    ops.bind_streams(torch.device('cuda', args.gpu))      # before RCCL takes its share of the hardware queues
    args.dist_backend = 'nccl'
    print('| distributed init (rank {}): {}'.format(args.rank, args.dist_url), flush=True)
    dist.init_process_group(backend=args.dist_backend, init_method=args.dist_url, world_size=args.world_size,
                            rank=args.rank, device_id=torch.device('cuda', args.gpu))
    dist.barrier()
