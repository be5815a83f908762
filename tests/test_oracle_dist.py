"""Oracle under 2-rank gloo: AllGather slice-only backward + DDP averaging vs the reference's own 2-rank run
(fixture model_tiny_w2.npz, captured from the reference under DistributedDataParallel)."""
import os

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.helpers import T, grad_close, load_npz, weights

TINY_CFG = {'weight_reg_loss': 0.01, 'weight_caption_loss': 0.01, 'weight_dstl_loss': 0.5, 'weight_cv_loss': 0.01}


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(2)
    from oracle import leccr_oracle as O
    z1, z2 = load_npz('model_tiny_w1.npz'), load_npz('model_tiny_w2.npz')
    sd = weights(z1, requires_grad=True)
    b = 8 // world
    sl = slice(rank * b, (rank + 1) * b)
    g = lambda k: T(z1[k])[sl]
    losses = O.retrieval_forward(sd, TINY_CFG, g('image'), g('ids_s'), g('mask_s'), g('ids_t'), g('mask_t'),
                                 g('ids_c'), g('mask_c'), g('idx'), text_heads=2)
    sum(losses).backward()
    bad = []
    got = np.array([float(l.detach()) for l in losses])
    if np.max(np.abs(got - z2[f'r{rank}/losses'])) > 1e-5:
        bad.append(('losses', got.tolist()))
    for k, v in z2.items():
        if k.startswith('r0/g/'):
            name = k[5:]
            gr = sd[name].grad.clone()
            dist.all_reduce(gr)            # DDP: SUM then / W  (image_Retrieval_caption.py:390-392)
            gr /= world
            if not grad_close(gr, v, 5e-4):
                bad.append(name)
    ret[rank] = bad
    dist.barrier()
    dist.destroy_process_group()


def test_oracle_two_ranks_match_reference_ddp():
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, 29631, ret), nprocs=2, join=True)
    assert ret[0] == [] and ret[1] == [], (ret[0], ret[1])
