"""Data-parallel semantics on the GPU box: 2 ranks (gloo rendezvous, both on cuda:0) run the HIP-backed model on the
two halves of the fixture batch; losses per rank and the rank-averaged parameter gradients must equal what the
REFERENCE produced under 2-rank DistributedDataParallel (tests/golden/model_tiny_w2.npz) - including the mixed
1/W-vs-full scaling of parameters that sit after the all-gather (temp, cproj, vproj)."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


_managers = []


def _manager():
    """An mp.Manager whose server process is SPAWNED (a forked one inherits this process's open GPU device and counts against the
    GPU box's limit of 6 processes on the card for as long as it lives); earlier tests' managers are shut down first."""
    while _managers:
        _managers.pop().shutdown()
    m = mp.get_context('spawn').Manager()
    _managers.append(m)
    return m


def _init(rank, world, port):
    """Rendezvous of one test rank.  Default: gloo, every rank on cuda:0 (the one-GPU development box).  With
    LECCR_TEST_BACKEND=nccl (tests/test_rccl_gpu.py, nodes with >= 2 GPUs): RCCL, one rank per device - the production layout."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    backend = os.environ.get('LECCR_TEST_BACKEND', 'gloo')
    if backend == 'nccl':
        from leccr_amd.parallel import setup_rccl_env
        setup_rccl_env()
        torch.cuda.set_device(rank)
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', rank))
    else:
        dist.init_process_group('gloo', rank=rank, world_size=world)
        torch.cuda.set_device(0)


def _worker(rank, world, port, ret):
    _init(rank, world, port)
    from leccr_amd.optim import create_optimizer
    from leccr_amd.parallel import attach_grad_sync
    from tests.helpers import grad_close, load_npz, weights
    from tests.test_model_gpu import MODEL_VIT, TINY_BERT, build, run
    z1, z2 = load_npz('model_tiny_w1.npz'), load_npz(f'model_tiny_w{world}.npz')
    model = build(dict(clip_geometry=MODEL_VIT, text_config=TINY_BERT, compute_dtype='fp32'), weights(z1))
    opt = attach_grad_sync(create_optimizer(dict(lr=0.0, weight_decay=0.0), model))
    b = 8 // world
    losses = run(model, z1, slice(rank * b, (rank + 1) * b))
    opt.zero_grad()
    sum(losses).backward()
    opt.grad_sync()                                   # SUM over ranks; the 1/W rides in the AdamW launch (grad_scale)
    bad = []
    got = np.array([float(l.detach()) for l in losses])
    if np.max(np.abs(got - z2[f'r{rank}/losses'])) > 1e-4:
        bad.append(('losses', got.tolist(), z2[f'r{rank}/losses'].tolist()))
    named = dict(model.named_parameters())
    n = 0
    for k, g in z2.items():
        if k.startswith('r0/g/'):
            n += 1
            mine = named[k[5:]].grad * opt.grad_scale
            if not grad_close(mine, g, 2e-3, atol=2e-6):
                bad.append((k[5:], float((mine.cpu() - torch.from_numpy(g)).abs().max()), float(np.abs(g).max())))
    ret[rank] = (n, bad[:6])
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_losses_and_gradients_match_reference_ddp():
    mgr = _manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, 29651, ret), nprocs=2, join=True)
    for r in (0, 1):
        n, bad = ret[r]
        assert n > 100 and not bad, (r, bad)


def test_four_rank_losses_and_gradients_match_reference_ddp():
    """The reference captured under FOUR-rank DDP (tests/golden/model_tiny_w4.npz; SURVEY.md 8(c) `losses_w4`): four HIP ranks on the
    GPU reproduce its per-rank losses and rank-averaged gradients - the packed gather, the local-rows backward and the 1/W-vs-full
    scaling (models/xvlm.py:50-70, model_retrieval_caption.py:94-143) at W = 4, against the reference itself, not the oracle."""
    mgr = _manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(4, 29655, ret), nprocs=4, join=True)
    for r in range(4):
        n, bad = ret[r]
        assert n > 100 and not bad, (r, bad)


def _train_worker(rank, world, port, fused, ret):
    _init(rank, world, port)
    from leccr_amd import ops
    from leccr_amd.optim import create_optimizer
    from leccr_amd.parallel import attach_grad_sync
    from tests.helpers import load_npz, weights
    from tests.test_model_gpu import MODEL_VIT, TINY_BERT, build, run
    z1 = load_npz('model_tiny_w1.npz')
    model = build(dict(clip_geometry=MODEL_VIT, text_config=TINY_BERT, compute_dtype='fp32'), weights(z1))
    opt = attach_grad_sync(create_optimizer(dict(lr=1e-3, weight_decay=0.01), model))
    if fused:
        opt.enable_backward_fusion()
    b = 8 // world
    for _ in range(2):
        losses = run(model, z1, slice(rank * b, (rank + 1) * b))
        opt.zero_grad()
        sum(losses).backward()
        opt.step()
    a = model.arena
    ret[(fused, rank)] = (a.data.detach().cpu().numpy().copy(), a.step_count)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_training_with_the_optimizer_fused_into_backward():
    """Two optimizer steps on 2 ranks: every all-reduce phase (early region during the ViT backward, ViT thirds, rest at step())
    and the early parameter update behind its all-reduce give both ranks the same weights, equal to the plain optimizer's."""
    mgr = _manager()
    ret = mgr.dict()
    mp.spawn(_train_worker, args=(2, 29671, True, ret), nprocs=2, join=True)
    mp.spawn(_train_worker, args=(2, 29673, False, ret), nprocs=2, join=True)
    (f0, n0), (f1, _), (p0, _), (p1, _) = ret[(True, 0)], ret[(True, 1)], ret[(False, 0)], ret[(False, 1)]
    assert n0 == 2
    assert np.array_equal(f0, f1) and np.array_equal(p0, p1)               # ranks stay in lock step (same summed gradients)
    # fused = plain: the gradient atomics' order differs, and Adam's m / sqrt(v) amplifies that on near-zero gradients (lr 1e-3: an element
    # whose gradient is rounding noise moves by up to 2e-3 over two steps).  Round 4's fp32 path sums more tensors with atomics (the bias
    # gradients ride in the split passes): measured 0 - 1.5e-5 from run to run where it was <= 1e-5 - a schedule race would show 1e-3 and up
    assert np.max(np.abs(f0 - p0)) <= 5e-5 * np.max(np.abs(p0))


def _eval_worker(rank, world, port, ret):
    import types
    _init(rank, world, port)
    from leccr_amd.evaluation import evaluation_coarse
    from tests.helpers import load_npz, weights
    from tests.test_model_gpu import MODEL_VIT, TINY_BERT, build
    z = load_npz('model_tiny_w1.npz')
    model = build(dict(clip_geometry=MODEL_VIT, text_config=TINY_BERT, compute_dtype='fp32'), weights(z))
    ids_s, m_s = torch.from_numpy(z['ids_s']), torch.from_numpy(z['mask_s'])
    ids_c, m_c = torch.from_numpy(z['ids_c']), torch.from_numpy(z['mask_c'])
    image = torch.from_numpy(z['image'])

    class Tok:
        def __call__(self, text, **kw):
            ids, m = (ids_s, m_s) if text[0][0] == 't' else (ids_c, m_c)
            sel = torch.tensor([int(t[1:]) for t in text])
            o = types.SimpleNamespace(input_ids=ids[sel], attention_mask=m[sel])
            o.to = lambda dev: types.SimpleNamespace(input_ids=o.input_ids.to(dev), attention_mask=o.attention_mask.to(dev))
            return o

    class Loader(list):
        dataset = types.SimpleNamespace(text=[f't{i}' for i in range(8)])
    # 3 image batches (3 + 3 + 2) and 3 text chunks (3 + 3 + 2): ragged shares per rank
    loader = Loader([(image[a:b], [f'c{j}' for j in range(a, b)], None) for a, b in ((0, 3), (3, 6), (6, 8))])
    # (groups of 3 images / 3 texts per forward, so that both ranks get ragged shares; the default regroups to 64 / 256)
    cfg = {'batch_size_test_text': 3, 'max_tokens': 200, 'eval_group_images': 3, 'eval_group_texts': 3}
    i2t, t2i = evaluation_coarse(model, loader, Tok(), 'cuda', cfg)                 # sharded over the 2 ranks
    full, _ = evaluation_coarse(model, loader, Tok(), 'cuda', cfg, shard=False)     # every rank encodes everything
    ret[rank] = (i2t.shape, float(np.max(np.abs(i2t - full))), bool(np.array_equal(t2i, i2t.T)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_evaluation_matches_unsharded():
    mgr = _manager()
    ret = mgr.dict()
    mp.spawn(_eval_worker, args=(2, 29653, ret), nprocs=2, join=True)
    for r in (0, 1):
        shape, err, tr = ret[r]
        assert shape == (8, 8) and err <= 1e-6 and tr, (r, shape, err, tr)
