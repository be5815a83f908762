"""Data-parallel plumbing: one process per GPU, torch.distributed over RCCL ('nccl' backend on ROCm) / xGMI.

The reference wraps the model in torch DistributedDataParallel (image_Retrieval_caption.py:390-392).  Here parameter
gradients live in ONE flat f32 arena, so the gradient exchange is a few large all-reduces over contiguous memory
(mean over ranks = SUM then the 1/W folded into the fused AdamW launch), issued from optimizer.step().
"""
import ctypes
import os

import torch
import torch.distributed as dist

# LECCR_EMULATE_RCCL="blocks:us_per_GB" (diagnostic, one-rank worlds): after each phased all-reduce, `blocks` workgroups hold their CUs for
# us_per_GB x the gigabytes reduced - what an RCCL kernel of that many channels does to the GEMMs beside it (scripts/r5_rccl_cus.sh)
_EMULATE = tuple(int(x) for x in os.environ['LECCR_EMULATE_RCCL'].split(':')) if os.environ.get('LECCR_EMULATE_RCCL') else None
_BUCKET = 64 * 1024 * 1024      # elements per all-reduce (256 MiB f32): large, few collectives (xGMI per-link bound)


def rccl_cus():
    """CUs reserved for RCCL while an all-reduce overlaps the backward pass (LECCR_RCCL_CUS, default 0 = off).

    The persistent GEMMs run one workgroup per CU with a static item list and need the whole CU (160 KiB LDS, every VGPR): a
    collective's workgroups resident on n CUs would push n of a 256-workgroup launch into a second round.  With LECCR_RCCL_CUS=n
    RCCL is capped to n channels (NCCL_MAX_NCHANNELS, unless the user set it) and the GEMM launches enqueued while an all-reduce
    of the step is in flight use 256 - n CUs.  It is OFF by default: the trade has not been measured on a multi-GPU node yet
    (this project's development box has one GPU), and an unmeasured cap must not cost the single-GPU path anything."""
    try:
        return max(0, min(int(os.environ.get('LECCR_RCCL_CUS', '0')), 128))
    except ValueError:
        return 0


def setup_rccl_env():
    """Environment RCCL reads at communicator creation - called by BOTH init paths (this module's and
    leccr_amd.utils.init_distributed_mode, the one the task scripts use) before init_process_group."""
    n = rccl_cus()
    if n > 0:
        os.environ.setdefault('NCCL_MAX_NCHANNELS', str(n))
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')         # dmabuf IPC (the host driver supports only that)


def init_distributed_mode(backend=None):
    """utils/__init__.py:306-328 without the argparse object: env:// rendezvous, one rank per GPU."""
    if 'RANK' not in os.environ or 'WORLD_SIZE' not in os.environ:
        return 0, 1, 0
    rank, world, local = int(os.environ['RANK']), int(os.environ['WORLD_SIZE']), int(os.environ.get('LOCAL_RANK', 0))
    setup_rccl_env()
    if torch.cuda.is_available():
        torch.cuda.set_device(local % torch.cuda.device_count())     # (ranks outnumber GPUs only in gloo rehearsals)
        from . import ops
        ops.bind_streams(torch.device('cuda', torch.cuda.current_device()))     # before RCCL takes its share of the hardware queues
    if not dist.is_initialized():
        backend = backend or ('nccl' if torch.cuda.is_available() else 'gloo')
        # RCCL: the rank's device is named up front (no "guessing device ID based on global rank" at the first barrier)
        dev = {'device_id': torch.device('cuda', torch.cuda.current_device())} if backend == 'nccl' else {}
        dist.init_process_group(backend=backend, init_method='env://', world_size=world, rank=rank, **dev)
        dist.barrier()
    return rank, world, local


def one_rank_world():
    """LECCR_ONE_RANK_WORLD=1: a process group of ONE rank takes the collective path all the same (packed all-gather, phased all-reduce on the
    reduce stream, sharded-evaluation gathers).  A rehearsal aid for a one-GPU box: every RCCL call of the N > 1 path is issued for real
    (tests/test_dist_gpu.py::test_one_rank_rccl_world_*), on a communicator that has nobody to talk to."""
    return os.environ.get('LECCR_ONE_RANK_WORLD') == '1' and dist.is_available() and dist.is_initialized()


def collective_path(world):
    return world > 1 or one_rank_world()


def attach_grad_sync(optimizer):
    """Make optimizer.step() average the arena's gradients over ranks first (DDP semantics).

    Phases: the arena's early region (text tower, heads, interaction: ~70 % of the bytes) is all-reduced as soon as
    the ViT backward starts (ops.on_late_backward_start fires in the first ViT backward node, when every other gradient
    is final) and overlaps it on RCCL's stream; the ViT region follows in thirds as the block backwards complete
    (ops.on_vit_block_backward), the last third at optimizer.step()."""
    if not (dist.is_available() and dist.is_initialized()) or not collective_path(dist.get_world_size()):
        return optimizer
    from . import diag, ops
    world = dist.get_world_size()
    arena = optimizer.arena
    grad, split = arena.grad, arena.early_numel
    pending = []

    # RCCL on device tensors: the collectives are issued as SYNCHRONOUS ops, which c10d enqueues on the CURRENT stream (no internal stream
    # of RCCL's process group, no handles: torch >= 2.8), and that stream is the caption tower's - idle for the whole backward pass, with a
    # hardware queue of its own (ops.bind_streams).  What the rest of the code waits for is an event on it.  gloo (CPU tests, the
    # one-GPU rehearsals): async handles as before.
    on_stream = grad.is_cuda and dist.get_backend() == 'nccl'

    class _Done:
        def __init__(self, stream):
            self.ev = torch.cuda.Event()
            self.ev.record(stream)

        def wait(self):
            torch.cuda.current_stream().wait_event(self.ev)

    def reduce_range(lo, hi):
        for o in range(lo, hi, _BUCKET):
            w = dist.all_reduce(grad[o:min(o + _BUCKET, hi)], async_op=not on_stream)
            if not on_stream:
                pending.append(w)
        if on_stream and hi > lo:
            pending.append(_Done(torch.cuda.current_stream()))

    reduce_stream = []
    late = {'hi': grad.numel()}               # [late['hi'], numel) of the ViT region is already handed to RCCL
    cus = rccl_cus() if grad.is_cuda and dist.get_backend() == 'nccl' else 0

    def gemm_cus(n):
        if cus:
            from . import _lib
            _lib.lib().leccr_set_gemm_max_blocks(n)

    def reduce_async(lo, hi):
        """All launches that write grad[lo:hi] are enqueued (on the main, text and weight-gradient streams): the reduce stream
        waits for the three and takes the all-reduces, so the main stream goes straight on with the backward pass."""
        if grad.is_cuda:
            gemm_cus(256 - cus)     # from here until sync(): GEMM launches leave RCCL its CUs (no-op unless LECCR_RCCL_CUS > 0)
            if not reduce_stream:
                reduce_stream.append((ops.caption_stream(grad.device) if on_stream else None) or torch.cuda.Stream(device=grad.device))
            rs = reduce_stream[0]
            rs.wait_stream(torch.cuda.current_stream())
            ops.join_wgrad_stream(rs)
            with torch.cuda.stream(rs):
                reduce_range(lo, hi)
                if _EMULATE:            # (diagnostic) stand-in for the collective's resident workgroups: blocks:us per GB reduced
                    from . import _lib
                    us = int(_EMULATE[1] * (hi - lo) * 4 / 1e9)
                    _lib.lib().leccr_debug_occupy(_EMULATE[0], 256, us, ctypes.c_void_p(rs.cuda_stream))
                    pending.append(_Done(rs))
        else:
            reduce_range(lo, hi)

    def early():
        # first ViT backward node: autograd has run the younger text-tower / interaction nodes, the early region is final
        if not pending:                       # once per step
            with diag.host('reduce_early'):
                reduce_async(0, split)
            optimizer._early_handles = list(pending)     # a fused early update (optim.py) waits for exactly these

    def vit_block(blk):
        # start of a ViT block's backward: the blocks behind it in the arena (and ln_post / proj) are final.  Their share of
        # the ViT region goes out as soon as it is a third of the region, so only the first blocks' gradients are left for
        # optimizer.step() (unbucketed, the ViT region's 344 MB all-reduce sat exposed between backward and AdamW).
        end = max(arena.offsets[p._aidx] + p.numel() for p in blk.parameters())
        end = min((end + 63) // 64 * 64, late['hi'])
        if late['hi'] - end >= (grad.numel() - split) // 3:
            with diag.host('reduce_vit_third'):
                reduce_async(end, late['hi'])
            late['hi'] = end

    def sync():
        with diag.host('grad_sync'):
            _sync()

    def _sync():
        ops.join_wgrad_stream()
        if not pending:                       # no ViT backward ran (e.g. a text-only step): reduce everything now
            reduce_range(0, split)
        reduce_range(split, late['hi'])
        late['hi'] = grad.numel()
        diag.begin('allreduce_exposed')          # how long the stream of optimizer.step() waits here = what the backward did not hide
        for w in pending:
            w.wait()
        diag.end('allreduce_exposed')
        pending.clear()
        gemm_cus(0)

    phased = split < grad.numel() and os.environ.get('LECCR_DDP_PHASES', '1') != '0'      # 0: one exchange at optimizer.step() (A/B)
    arena.on_late_backward_start = early if phased else None       # this arena's backward only (ops._fire)
    arena.on_vit_block_backward = vit_block if phased else None
    optimizer.grad_sync = sync
    optimizer.grad_scale = 1.0 / world
    return optimizer
