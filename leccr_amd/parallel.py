"""Data-parallel plumbing: one process per GPU, torch.distributed over RCCL ('nccl' backend on ROCm) / xGMI.

The reference wraps the model in torch DistributedDataParallel (image_Retrieval_caption.py:390-392).  Here parameter
gradients live in ONE flat f32 arena, so the gradient exchange is a few large all-reduces over contiguous memory
(mean over ranks = SUM then the 1/W folded into the fused AdamW launch), issued from optimizer.step().
"""
import os

import torch
import torch.distributed as dist

_BUCKET = 64 * 1024 * 1024      # elements per all-reduce (256 MiB f32): large, few collectives (xGMI per-link bound)


def init_distributed_mode(backend=None):
    """utils/__init__.py:306-328 without the argparse object: env:// rendezvous, one rank per GPU."""
    if 'RANK' not in os.environ or 'WORLD_SIZE' not in os.environ:
        return 0, 1, 0
    rank, world, local = int(os.environ['RANK']), int(os.environ['WORLD_SIZE']), int(os.environ.get('LOCAL_RANK', 0))
    if torch.cuda.is_available():
        torch.cuda.set_device(local)
    if not dist.is_initialized():
        dist.init_process_group(backend=backend or ('nccl' if torch.cuda.is_available() else 'gloo'),
                                init_method='env://', world_size=world, rank=rank)
        dist.barrier()
    return rank, world, local


def attach_grad_sync(optimizer):
    """Make optimizer.step() average the arena's gradients over ranks first (DDP semantics)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return optimizer
    world = dist.get_world_size()
    grad = optimizer.arena.grad

    def sync():
        works = [dist.all_reduce(grad[o:o + _BUCKET], async_op=True) for o in range(0, grad.numel(), _BUCKET)]
        for w in works:
            w.wait()

    optimizer.grad_sync = sync
    optimizer.grad_scale = 1.0 / world
    return optimizer
