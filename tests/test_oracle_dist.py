"""Oracle under 2- and 4-rank gloo: AllGather slice-only backward + DDP averaging vs the reference's own 2- and 4-rank runs
(fixtures model_tiny_w2.npz / model_tiny_w4.npz, captured from the reference under DistributedDataParallel)."""
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
    z1, z2 = load_npz('model_tiny_w1.npz'), load_npz(f'model_tiny_w{world}.npz')
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


def test_oracle_four_ranks_match_reference_ddp():
    """SURVEY.md 8(c) `losses_w4`: the reference itself captured at W = 4 (tests/golden/make_golden.py w4) - the oracle under 4-rank
    gloo reproduces its per-rank losses and DDP-averaged gradients (1/4 for pre-gather parameters, full for temp / cproj / vproj)."""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(4, 29635, ret), nprocs=4, join=True)
    assert all(ret[r] == [] for r in range(4)), [ret[r] for r in range(4)]


def test_world_fixtures_are_the_oracle_under_w_rank_gloo():
    """tests/golden/world_oracle_w*_g*.npz (what tests/test_world_gpu.py holds the HIP path to at the world shapes of BASELINE configs C
    and E) really are the oracle run under W-rank gloo: re-run live here at W = 4 and W = 8, G = 1024 (eight CPU processes), compared
    with what the fixtures keep.  Also the invariant SURVEY.md section 4 names: every rank reports the same G x G losses (the two
    local-batch terms differ per rank), and they equal the single-process losses on the concatenated batch."""
    from tests.test_world_gpu import WL, grad_sample, load_world_fixture, oracle_world, world_batch, world_model
    for world in (4, 8):
        losses, grads = oracle_world(world, 1024)
        z = load_world_fixture(world, 1024)
        assert np.max(np.abs(np.asarray(losses) - z['losses'])) < 1e-5
        names = [k[2:] for k in z if k.startswith('n/')]
        assert set(names) == set(grads) and len(names) > 100
        for k in names:
            wn = float(z['n/' + k])
            assert abs(float(np.linalg.norm(grads[k].astype(np.float64))) - wn) <= 1e-4 * wn + 1e-9, k
            assert np.max(np.abs(grad_sample(grads[k]) - z['s/' + k])) <= 1e-4 * max(np.max(np.abs(z['s/' + k])), 1e-7) + 1e-8, k
        lo = np.asarray(losses)
        assert np.max(np.abs(lo[:, :3] - lo[0, :3])) < 1e-5          # the three gathered terms: identical on every rank
    # W = 1 on the whole batch gives the same gathered-loss values (dropout off)
    from oracle import leccr_oracle as O
    sd = {k: v.detach() for k, v in world_model().state_dict().items() if not k.startswith('caption_encoder.')}
    zb = world_batch(1024)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29899')
    dist.init_process_group('gloo', rank=0, world_size=1)
    try:
        with torch.no_grad():
            one = O.retrieval_forward(sd, WL, zb['image'], zb['ids_s'], zb['mask_s'], zb['ids_t'], zb['mask_t'], zb['ids_c'], zb['mask_c'],
                                      zb['idx'], text_heads=2)
    finally:
        dist.destroy_process_group()
    assert abs(float(one[2]) - float(lo[0, 2])) < 1e-4, (float(one[2]), lo[0, 2])
