"""The data-parallel path over RCCL ('nccl' backend), one rank per GPU - runs only where the node has >= 2 GPUs (the
development box has one: there the same workers run under gloo in tests/test_dist_gpu.py).  Covers what no gloo test can:
`all_gather_into_tensor` in AllGather / AllGatherPacked (models/xvlm.py:50-70 semantics), the phased all-reduce on the reduce
stream behind the side streams, the CU reservation for RCCL (LECCR_RCCL_CUS), and bench.py's self-launch."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from tests import test_dist_gpu as D

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs >= 2 GPUs (RCCL, one rank per device)')]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(autouse=True)
def _nccl_backend(monkeypatch):
    monkeypatch.setenv('LECCR_TEST_BACKEND', 'nccl')
    monkeypatch.setenv('HSA_ENABLE_IPC_MODE_LEGACY', '0')


def test_rccl_two_rank_losses_and_gradients_match_reference_ddp():
    """AllGatherPacked / AllGather over RCCL + the summed gradients: the reference's 2-rank DDP fixture, as under gloo."""
    ret = D._manager().dict()
    mp.spawn(D._worker, args=(2, 29751, ret), nprocs=2, join=True)
    for r in (0, 1):
        n, bad = ret[r]
        assert n > 100 and not bad, (r, bad)


@pytest.mark.parametrize('cus', ['0', '16'])
def test_rccl_phased_all_reduce_equals_plain_and_cu_reservation_changes_nothing(cus, monkeypatch):
    """Two optimizer steps: early region reduced on the reduce stream during the ViT backward + ViT thirds + rest at step(), with
    the optimizer fused into backward, against the plain optimizer; LECCR_RCCL_CUS = 16 (GEMMs on 240 CUs while a reduce is in
    flight, NCCL_MAX_NCHANNELS = 16) must give the same weights as 0."""
    monkeypatch.setenv('LECCR_RCCL_CUS', cus)
    ret = D._manager().dict()
    mp.spawn(D._train_worker, args=(2, 29761 + int(cus), True, ret), nprocs=2, join=True)
    mp.spawn(D._train_worker, args=(2, 29781 + int(cus), False, ret), nprocs=2, join=True)
    (f0, n0), (f1, _), (p0, _), (p1, _) = ret[(True, 0)], ret[(True, 1)], ret[(False, 0)], ret[(False, 1)]
    assert n0 == 2 and np.array_equal(f0, f1) and np.array_equal(p0, p1)
    assert np.max(np.abs(f0 - p0)) <= 1e-5 * np.max(np.abs(p0))


def test_rccl_sharded_evaluation_matches_unsharded():
    ret = D._manager().dict()
    mp.spawn(D._eval_worker, args=(2, 29753, ret), nprocs=2, join=True)
    for r in (0, 1):
        shape, err, tr = ret[r]
        assert shape == (8, 8) and err <= 1e-6 and tr, (r, shape, err, tr)


def test_bench_self_launch_two_gpus():
    """`python bench.py --gpus 2` with no launcher starts its own two ranks over RCCL and reports n_gpus = 2."""
    env = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'LECCR_BENCH_BACKEND'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--batch', '32', '--steps', '2', '--warmup', '1',
                        '--roofline-steps', '0'], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out['n_gpus'] == 2 and out['config']['backend'] == 'nccl' and out['config']['global_batch'] == 64
    assert out['value'] > 0 and out['config']['rccl_version']


@pytest.mark.parametrize('world,G', [(4, 1024), (8, 1024), (8, 2048)])
def test_rccl_world_shapes_of_configs_c_and_e_match_the_oracle_world(world, G):
    """BASELINE configs C / E world shapes over RCCL, one rank per GPU (W = 4 and W = 8, global batch 1024 / 2048, tiny towers, production
    loss stage): per-rank losses and rank-averaged gradients against the oracle run under W-rank gloo (tests/golden/world_oracle_*.npz) -
    the same check tests/test_world_gpu.py makes with W = 4 gloo ranks on one GPU and with the 8-rank loopback world.  Lights up on a
    node with >= W GPUs (the first 8-GPU lease), skipped below."""
    if torch.cuda.device_count() < world:
        pytest.skip(f'needs >= {world} GPUs')
    import tempfile
    from tests import test_world_gpu as Wd
    z = Wd.load_world_fixture(world, G)
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(Wd._hip_rank, args=(world, 29871 + world + G // 1024, G, d), nprocs=world, join=True)
        res = [torch.load(os.path.join(d, f'hip_{r}.pt'), weights_only=False) for r in range(world)]
    n, bad = Wd._check([r['losses'] for r in res], res[0]['grads'], z)
    assert n > 60 and not bad, (n, bad[:6])
