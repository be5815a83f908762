"""Host-side logic that needs no GPU: reference-compatible module tree / state_dict keys, optimizer grouping rule,
LR schedule, synthetic batches, AllGather autograd semantics and gradient averaging under 2-rank gloo."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.helpers import load_npz

TINY = {'embed_dim': 16, 'temp': 0.07, 'use_one_cl_proj_only': False, 'caption_encoder_name': 'mbert',
        'num_queries': 4, 'caption_ca_layer': 2, 'caption_interaction_layer': 2, 'weight_reg_loss': 0.01,
        'weight_caption_loss': 0.01, 'weight_dstl_loss': 0.5, 'weight_cv_loss': 0.01,
        'clip_geometry': dict(embed_dim=32, image_resolution=64, vision_layers=2, vision_width=64,
                              vision_patch_size=16, context_length=8, vocab_size=32, transformer_width=64,
                              transformer_heads=1, transformer_layers=1),
        'text_config': dict(vocab_size=100, hidden_size=64, num_hidden_layers=2, num_attention_heads=2,
                            intermediate_size=128, max_position_embeddings=40)}


def _model():
    from leccr_amd.models import RetrievalModel
    m = RetrievalModel(dict(TINY))
    m.init_caption_encoder()
    return m


def test_state_dict_keys_and_shapes_match_the_reference_fixture():
    z = load_npz('model_tiny_w1.npz')
    sd = _model().state_dict()
    ref = {k[2:]: v.shape for k, v in z.items() if k.startswith('w/')}
    assert set(ref) <= set(sd)
    for k, shp in ref.items():
        assert tuple(sd[k].shape) == tuple(shp), k
    extra = [k for k in sd if k not in ref and not k.startswith('caption_encoder.')]
    assert not extra, extra[:5]
    # the aliased caption encoder appears under both prefixes, as in the reference (model_retrieval_caption.py:35)
    assert 'caption_encoder.embeddings.word_embeddings.weight' in sd


def test_unused_parameters_are_the_references():
    from leccr_amd.models.xvlm import is_unused_parameter
    names = [n for n, _ in _model().named_parameters()]
    unused = {n for n in names if is_unused_parameter(n)}
    z = load_npz('model_tiny_w1.npz')
    no_grad_in_reference = {k[2:] for k in z if k.startswith('w/')} - {k[2:] for k in z if k.startswith('g/')}
    no_grad_in_reference = {n for n in no_grad_in_reference if n in names}
    assert unused == no_grad_in_reference, sorted(unused ^ no_grad_in_reference)[:6]


def test_forward_on_cpu_fails_loudly():
    m = _model()
    with pytest.raises(RuntimeError, match='MI355X'):
        m.get_vision_embeds(torch.zeros(1, 3, 64, 64))


def test_decay_rule_matches_optim_py():
    from leccr_amd.optim import decay_of_factory
    d = decay_of_factory(0.01)
    assert d('text_encoder.encoder.layer.0.output.LayerNorm.weight', None) == 0.0
    assert d('crossattn.layers.0.norm2.weight', None) == 0.0
    assert d('vision_proj.bias', None) == 0.0
    assert d('clip_encoder.visual.ln_pre.weight', None) == 0.01        # 'ln_pre.weight' matches no no-decay substring
    assert d('temp', None) == 0.01 and d('queries', None) == 0.01


def test_scheduler_is_linear_warmup_then_decay():
    from leccr_amd.scheduler import create_scheduler
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([p], lr=1.0)
    sched = create_scheduler(dict(sched='linear', epochs=2, step_per_epoch=10, num_warmup_steps=0.1), opt)
    lrs = []
    for _ in range(20):
        lrs.append(opt.param_groups[0]['lr'])
        opt.step()
        sched.step()
    assert lrs[0] == 0.0 and lrs[1] == 0.5 and lrs[2] == 1.0
    assert abs(lrs[11] - (20 - 11) / 18) < 1e-12 and lrs[-1] > 0


def test_synthetic_batch_contract():
    from leccr_amd.synthetic import CLS, SEP, make_batch
    b = make_batch(32, rank=1)
    assert b['image'].shape == (32, 3, 384, 384) and b['image'].dtype == torch.float32
    for enc, L in ((b['text'][0], 32), (b['text'][1], 32), (b['captions'], 64)):
        ids, m = enc.input_ids, enc.attention_mask
        assert ids.shape == (32, L) and ids.dtype == torch.int64
        assert (ids[:, 0] == CLS).all() and ((ids == SEP).sum(1) == 1).all()
        assert ((ids == 0) == (m == 0)).all() and m[:, 0].all()
    assert b['idx'][15] == b['idx'][14] and b['idx'][0] == 32             # 1-in-16 duplicates, rank offset
    assert torch.equal(make_batch(4)['image'], make_batch(4)['image'])    # seeded


def _gather_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from leccr_amd.models.xvlm import allgather, gather_rows
    from leccr_amd.parallel import attach_grad_sync
    x = (torch.arange(6, dtype=torch.float32).view(3, 2) + 10 * rank).requires_grad_(True)
    g = allgather(x, rank, world)
    w = torch.arange(12, dtype=torch.float32).view(6, 2)
    (g * w).sum().backward()
    ok = torch.equal(g.detach()[3 * rank:3 * rank + 3], x.detach()) and g.shape == (6, 2)
    ok &= torch.equal(x.grad, w[3 * rank:3 * rank + 3])                  # ONLY the local slice (xvlm.py:62-67)
    ok &= gather_rows(torch.tensor([[rank]])).view(-1).tolist() == [0, 1]
    # several blocks + the int64 idx column in ONE collective; gather_rows() then serves them from the step's cache
    from leccr_amd.models import xvlm
    xvlm.begin_step_gathers()
    p = (torch.arange(4, dtype=torch.float32).view(2, 2) + 100 * rank).requires_grad_(True)
    q = (torch.arange(6, dtype=torch.float32).view(2, 3) - 7 * rank).requires_grad_(True)
    ids = torch.tensor([[5 + rank], [2 ** 40 + rank]])                   # int64 bit patterns survive the f32 transport
    xvlm.gather_rows_packed([p, q], ids)
    n_cached = len(xvlm._gathered)
    gp, gq, gi = gather_rows(p), gather_rows(q), gather_rows(ids)
    ok &= n_cached == 3 and len(xvlm._gathered) == 3                     # no further collective
    ok &= torch.equal(gp.detach(), torch.cat([torch.arange(4.).view(2, 2), torch.arange(4.).view(2, 2) + 100]))
    ok &= torch.equal(gq.detach()[2 * rank:2 * rank + 2], q.detach()) and gq.shape == (4, 3)
    ok &= gi.view(-1).tolist() == [5, 2 ** 40, 6, 2 ** 40 + 1]
    wq = torch.arange(12, dtype=torch.float32).view(4, 3)
    ((gp * 2).sum() + (gq * wq).sum()).backward()
    ok &= torch.equal(p.grad, torch.full((2, 2), 2.0)) and torch.equal(q.grad, wq[2 * rank:2 * rank + 2])
    xvlm.begin_step_gathers()

    class FakeOpt:                                                       # what attach_grad_sync needs
        class arena:
            grad = torch.full((10,), float(rank + 1))
            early_numel = 6                                              # [0,6) early region, [6,10) late (ViT) region
            on_late_backward_start = on_vit_block_backward = None
        grad_sync, grad_scale = None, 1.0
    from leccr_amd import ops
    opt = attach_grad_sync(FakeOpt)
    FakeOpt.arena.on_late_backward_start()                               # what the first ViT backward node fires (ops._fire)
    opt.grad_sync()
    ok &= torch.allclose(FakeOpt.arena.grad * opt.grad_scale, torch.full((10,), 1.5))   # mean over the 2 ranks
    ret[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


def test_allgather_and_grad_sync_two_ranks_gloo():
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_gather_worker, args=(2, 29641, ret), nprocs=2, join=True)
    assert ret[0] and ret[1]


def _gather_worker_w(rank, world, port, ret):
    """World size > 2 (BASELINE configs C / D / E run at W = 4 and 8): packed gather layout, local-slice backward, arena mean."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(1)
    from leccr_amd.models import xvlm
    from leccr_amd.models.xvlm import gather_rows
    from leccr_amd.parallel import attach_grad_sync
    b = 3
    xvlm.begin_step_gathers()
    p = (torch.arange(b * 4, dtype=torch.float32).view(b, 4) + 1000 * rank).requires_grad_(True)
    q = (torch.arange(b * 2 * 5, dtype=torch.float32).view(b, 2, 5) - 50 * rank).requires_grad_(True)       # a [b, n, D] block
    ids = (torch.arange(b).view(b, 1) + b * rank + 2 ** 33)
    xvlm.gather_rows_packed([p, q], ids)
    gp, gq, gi = gather_rows(p), gather_rows(q), gather_rows(ids)
    ok = gp.shape == (world * b, 4) and gq.shape == (world * b, 2, 5)
    ok &= gi.view(-1).tolist() == [2 ** 33 + i for i in range(world * b)]
    for r in range(world):                                   # rank-major row order, as torch.cat(all_gather(...)) gives
        ok &= torch.equal(gp.detach()[r * b:(r + 1) * b], torch.arange(b * 4, dtype=torch.float32).view(b, 4) + 1000 * r)
        ok &= torch.equal(gq.detach()[r * b:(r + 1) * b], torch.arange(b * 2 * 5, dtype=torch.float32).view(b, 2, 5) - 50 * r)
    wp = torch.arange(world * b * 4, dtype=torch.float32).view(world * b, 4)
    ((gp * wp).sum() + gq.sum() * 3).backward()
    ok &= torch.equal(p.grad, wp[rank * b:(rank + 1) * b]) and torch.equal(q.grad, torch.full((b, 2, 5), 3.0))   # local slice only
    xvlm.begin_step_gathers()

    class FakeOpt:
        class arena:
            grad = torch.full((12,), float(rank + 1))
            early_numel = 8
            on_late_backward_start = on_vit_block_backward = None
        grad_sync, grad_scale = None, 1.0
    opt = attach_grad_sync(FakeOpt)
    FakeOpt.arena.on_late_backward_start()
    opt.grad_sync()
    ok &= torch.allclose(FakeOpt.arena.grad * opt.grad_scale, torch.full((12,), (world + 1) / 2))     # mean of 1..W
    ret[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [4, 8])
def test_packed_gather_and_grad_sync_at_larger_world_sizes_gloo(world):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_gather_worker_w, args=(world, 29700 + world, ret), nprocs=world, join=True)
    assert all(ret[r] for r in range(world)), dict(ret)


def test_launcher_and_script_cli_match_the_reference_flags():
    import run
    import image_Retrieval_caption as task
    a = run.build_parser().parse_args(['--task', 'itr_multi30k_caption', '--dist', 'gpu0', '--config', 'c.yaml',
                                       '--bs', '128', '--epoch', '70', '--checkpoint', 'null', '--output_dir', 'o'])
    cmd = run.build_command(a)
    assert 'torch.distributed.run' in cmd and '--nproc-per-node=1' in cmd and '--master-addr 127.0.0.1' in cmd
    assert 'image_Retrieval_caption.py' in cmd and '--bs 128' in cmd and '--epoch 70' in cmd and '--config c.yaml' in cmd
    t = task.build_parser().parse_args(['--config', 'c.yaml', '--bs', '128', '--epoch', '2', '--evaluate'])
    assert t.bs == 128 and t.evaluate and t.checkpoint == 'null' and t.distributed is True and t.seed == 42
    assert {'train', 'evaluation_coarse', 'itm_eval', 'main'} <= set(dir(task))
    # run.py:108-114,141-142 / run_video.sh:4: the video task goes to the double-sim script with the same flags
    v = run.build_parser().parse_args(['--dist', 'f2', '--task', 'itr_video_caption', '--config', 'configs/cclm-base-ft/Retrieval_msrvtt.yaml',
                                       '--output_dir', 'outputs/msrvtt/full', '--bs', '128', '--seed', '42', '--epoch', '50',
                                       '--checkpoint', 'null'])
    vcmd = run.build_command(v)
    assert 'video_Retrieval_caption_double_sim.py' in vcmd and '--nproc-per-node=2' in vcmd and 'CUDA_VISIBLE_DEVICES=0,1' in vcmd
    assert '--bs 128' in vcmd and '--epoch 50' in vcmd and '--config configs/cclm-base-ft/Retrieval_msrvtt.yaml' in vcmd
    with pytest.raises(NotImplementedError):                    # run.py:144-145
        run.build_command(run.build_parser().parse_args(['--task', 'itr_nope', '--dist', '1']))


def test_prefetcher_and_deferred_scalars_passthrough_on_cpu():
    """leccr_amd.pipeline: same batches in the same order; scalars come back one push late; flush drains."""
    from leccr_amd.pipeline import DeferredScalars, DevicePrefetcher
    from leccr_amd.synthetic import make_batch
    batches = [make_batch(2, res=32, L_t=8, L_c=8, vocab=100, seed=s) for s in range(3)]
    loader = [(b['image'], b['text'], b['captions'], b['idx'], None) for b in batches]
    out = list(DevicePrefetcher(loader, 'cpu'))
    assert len(out) == 3 and len(DevicePrefetcher(loader, 'cpu')) == 3
    for got, want in zip(out, loader):
        assert torch.equal(got[0], want[0]) and torch.equal(got[3], want[3]) and got[4] is None
        assert torch.equal(got[1][1].input_ids, want[1][1].input_ids) and type(got[2]) is type(want[2])
    assert list(DevicePrefetcher([], 'cpu')) == []
    d = DeferredScalars()
    assert d.push([torch.tensor(1.0), torch.tensor(2.0)]) is None
    assert d.push([torch.tensor(3.0), torch.tensor(4.0)]) == [1.0, 2.0]
    assert d.flush() == [3.0, 4.0] and d.flush() is None


def test_bigfile_reader_matches_the_reference_semantics(tmp_path):
    """leccr_amd.bigfile.BigFile on a generated store: de-duplication, unknown names dropped, row order, by-index reads."""
    from leccr_amd.bigfile import BigFile
    rows, dims = 37, 12
    feats = np.random.RandomState(3).randn(rows, dims).astype(np.float32)
    names = [f'video{i}_{i % 5}' for i in range(rows)]
    (tmp_path / 'shape.txt').write_text(f'{rows} {dims}\n')
    (tmp_path / 'id.txt').write_text(' '.join(names))
    feats.tofile(tmp_path / 'feature.bin')
    bf = BigFile(str(tmp_path))
    assert bf.shape() == [rows, dims]
    got_names, vecs = bf.read(['video30_0', 'video2_2', 'nope', 'video30_0', 'video17_2'])
    assert got_names == ['video2_2', 'video17_2', 'video30_0']                 # increasing row number, unique, known only
    assert np.array_equal(np.asarray(vecs, dtype=np.float32), feats[[2, 17, 30]])
    assert bf.read_one('video5_0') == feats[5].tolist()
    n2, arr = bf.read_array([36, 0, 36], isname=False)
    assert n2 == [names[0], names[36]] and arr.dtype == np.float32 and np.array_equal(arr, feats[[0, 36]])
    assert bf.read(['absent']) == ([], [])
    (tmp_path / 'shape.txt').write_text(f'{rows + 1} {dims}\n')
    with pytest.raises((AssertionError, ValueError)):
        BigFile(str(tmp_path))


def test_bench_refuses_to_report_a_multi_gpu_number_it_cannot_measure():
    """`python bench.py --gpus 8` with no launcher and fewer GPUs than asked: non-zero exit, no JSON line (round 1 printed a
    one-rank result with exit code 0 here).  With enough GPUs it starts the ranks itself (tests/test_rccl_gpu.py)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'LECCR_BENCH_BACKEND')}
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '8'], capture_output=True, text=True, env=env,
                       timeout=300)
    assert r.returncode != 0 and r.stdout.strip() == '' and 'refusing' in r.stderr


def test_caption_txt_store_reader_matches_the_reference_dataset_classes(tmp_path):
    """leccr_amd.captions on the store of tests/golden/caption_store.json (written out here from the fixture): the records, texts,
    id maps and ground-truth tables equal what the reference's re_train_dataset_caption / re_eval_dataset_caption produced on
    it (dataset/retrieval_dataset.py:30-264), Multi30K-style and mscoco-style, pre_caption included."""
    import json
    from leccr_amd.captions import EvalAnnotations, TrainAnnotations, get_video_id, pre_caption
    z = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'caption_store.json')))
    for raw, want in z['pre_caption'].items():
        assert pre_caption(raw, 30) == want
    with pytest.raises(ValueError):
        pre_caption(' ., ', 30)
    assert get_video_id('abc.jpg#enc#3') == 'abc' and get_video_id('v7.mp4#enc2zh#0') == 'v7' and get_video_id('c1#enc#0') == 'c1'
    d = str(tmp_path)
    for rel, text in z['files'].items():
        os.makedirs(os.path.dirname(os.path.join(d, rel)), exist_ok=True)
        open(os.path.join(d, rel), 'w').write(text)
    images = os.path.join(d, 'images')
    cfgs = {'m30k': (dict(root_dir=d, image_root=images, dataset='multi30k', generated_caption_dir=os.path.join(d, 'gen'),
                          generated_caption_type='text', train_file=['train_enc.txt', 'train_enc2de.txt']),
                     'val_de.txt', os.path.join(d, 'val_de_trans.txt')),
            'coco': (dict(root_dir=os.path.join(d, 'coco'), image_root=images, dataset='mscoco',
                          generated_caption_dir=os.path.join(d, 'coco/gen'), generated_caption_type='text',
                          train_file=['train_enc.txt', 'train_enc2ja.txt']), 'val_ja.txt', None)}
    for tag, (cfg, ann, trans) in cfgs.items():
        ref = z[tag]
        tr = TrainAnnotations(cfg, 30)
        assert len(tr) == ref['train_len'] and tr.languages == ref['languages'] and tr.img_ids == ref['img_ids']
        for i, (_, caps, gen, img_idx, cap_id) in enumerate(ref['train_records']):
            path, caps2, gen2, img_idx2, cap_id2 = tr.record(i)
            assert (caps2, gen2, img_idx2, cap_id2) == (caps, gen, img_idx, cap_id), (tag, i)
            assert path.startswith(images + '/') and path.endswith('.jpg')
        ev = EvalAnnotations(cfg, ann, 30, test_trans_file=trans, type='eval')
        assert ev.text == ref['eval_text'] and ev.image == ref['eval_image'] and ev.cap_ids == ref['eval_cap_ids']
        assert {str(k): v for k, v in ev.txt2img.items()} == ref['txt2img']
        assert {str(k): v for k, v in ev.img2txt.items()} == ref['img2txt']
        assert getattr(ev, 'text_trans', None) == ref['text_trans'] and len(ev) == len(ref['eval_records'])
        for i, (_, gen, idx) in enumerate(ref['eval_records']):
            path, gen2, idx2 = ev.record(i)
            assert (gen2, idx2) == (gen, idx) and path.startswith(images + '/')
    # the mscoco image path goes through img_id/image_ids.txt
    assert TrainAnnotations(cfgs['coco'][0]).record(0)[0].endswith('COCO_val2014_000000391895.jpg')


class _WsTok:
    """Whitespace "tokenizer" with the HF call signature the train loop uses; counts its calls in THIS process."""
    calls = 0

    def __call__(self, texts, padding=None, max_length=None, return_tensors=None):
        from leccr_amd.synthetic import Encoding
        assert padding == 'longest' and max_length == 200 and return_tensors == 'pt' and isinstance(texts, list)
        _WsTok.calls += 1
        rows = [[101] + [len(w) for w in t.split()] + [102] for t in texts]
        L = max(map(len, rows))
        ids = torch.tensor([r + [0] * (L - len(r)) for r in rows])
        return Encoding(ids, (ids != 0).long())


class _TextDS(torch.utils.data.Dataset):
    def __len__(self):
        return 6

    def __getitem__(self, i):
        return torch.full((3, 4, 4), float(i)), [f'a b c {"d " * i}', f'ein zwei {"x " * i}'], f'caption number {i}', i, f'img{i}#enc#0'


def test_tokenizing_collate_moves_the_tokenizer_calls_into_the_loader_workers():
    """pipeline.TokenizingCollate: a DataLoader with worker processes yields batches whose three text fields
    are already tokenised with the reference's arguments (image_Retrieval_caption.py:46-58); tokenize_or_pass then leaves them
    alone, and tokenises plain string lists exactly like the loop did.  Image layout (texts at position 1) and the video layout
    behind the reference's frame-padding collate_fn (position 2)."""
    from torch.utils.data import DataLoader
    from leccr_amd.pipeline import DevicePrefetcher, TokenizingCollate, tokenize_or_pass
    _WsTok.calls = 0
    loader = DataLoader(_TextDS(), batch_size=3, num_workers=2, collate_fn=TokenizingCollate(_WsTok(), 200), timeout=60)
    batches = list(DevicePrefetcher(loader, 'cpu'))
    assert len(batches) == 2 and _WsTok.calls == 0                               # every tokenizer call ran in a WORKER process
    image, text_ls, cap, idx, cap_ids = batches[1]
    assert image.shape == (3, 3, 4, 4) and idx.tolist() == [3, 4, 5] and cap_ids[0] == 'img3#enc#0'
    assert text_ls[0].input_ids.shape == (3, 2 + 3 + 5) and text_ls[0].input_ids[0].tolist()[:6] == [101, 1, 1, 1, 1, 1]
    assert text_ls[1].attention_mask.sum(1).tolist() == [2 + 2 + 3, 2 + 2 + 4, 2 + 2 + 5] and cap.input_ids[:, 1].tolist() == [7, 7, 7]
    assert tokenize_or_pass(_WsTok(), text_ls[0], 200, 'cpu') is text_ls[0] and _WsTok.calls == 0
    plain = tokenize_or_pass(_WsTok(), ['a bb', 'ccc'], 200, 'cpu')              # the reference's in-loop call, unchanged
    assert _WsTok.calls == 1 and plain.input_ids.tolist() == [[101, 1, 2, 102], [101, 3, 102, 0]]

    def video_collate(samples):                      # shape of retrieval_dataset_video.collate_fn's result
        frames, en, zh, gen, ids, caps = zip(*samples)
        return torch.stack(frames), torch.ones(len(frames), 2).bool(), [list(en), list(zh)], list(gen), torch.tensor(ids), list(caps)
    vb = TokenizingCollate(_WsTok(), 200, base_collate=video_collate, text_pos=2)(
        [(torch.zeros(2, 8), 'one two', 'yi er san', 'gen cap', 7, 'v7#enc#0'), (torch.ones(2, 8), 'three', 'si', 'gen', 9, 'v9#enc#0')])
    assert vb[0].shape == (2, 2, 8) and vb[1].dtype == torch.bool and vb[4].tolist() == [7, 9]
    assert vb[2][1].input_ids.tolist() == [[101, 2, 2, 3, 102], [101, 2, 102, 0, 0]] and vb[3].input_ids.shape == (2, 4)


def test_fp32_split_path_host_decisions():
    """Host side of the fp32-on-the-matrix-cores path (kernels.gemm / ops.X3), no GPU: which shapes take the 256 x 256 kernel's fused copy,
    that CPU tensors never enter the split path (the product path still fails loudly on them), that the segment count is 3 or 6, and
    that the stage timers record nothing while they are off."""
    from leccr_amd import diag, kernels as K, ops
    assert K.gemm256_takes(147712, 3072, 2304) and K.gemm256_takes(16384, 3072, 2304)           # config-B MLP shapes, ViT and BERT
    assert not K.gemm256_takes(1154, 768, 2304) and not K.gemm256_takes(147712, 3076, 2304) and not K.gemm256_takes(147712, 3072, 100)
    a, b = torch.zeros(512, 768), torch.zeros(768, 768)
    assert K._split_plan(a, b, False, False, None, False, None) is None                          # CPU tensors: no plan ..
    with pytest.raises(Exception):
        K.gemm(a, b)                                                                             # .. and no fallback either
    assert K.NS in (3, 6) and K.FORM_X == 0 and K.FORM_G == 1 and ops.X3 == 'x3'
    prev = K.set_f32_segments(6)
    assert K.NS == 6 and K.set_f32_segments(prev) == 6 and K.NS == prev
    with pytest.raises(AssertionError):
        K.set_f32_segments(4)
    assert not diag.enabled
    diag.begin('x'); diag.end('x')
    assert diag._spans == {} and diag._open == {}


def test_sharded_evaluation_gives_every_rank_of_a_node_a_group():
    """evaluation_coarse regroups the loader's batches to 256 images / 1024 texts per forward AND shards those groups over the ranks (group c ->
    rank c % world).  On the Multi30K / MSCOCO-1k test sets (1000 images, 1000 - 5000 texts; image_Retrieval_caption.py:83-163) that left 4 image
    groups and 1 - 5 text chunks for 8 ranks (round-4 advisor finding): the group is capped at a 1 / world share of the set."""
    from leccr_amd.evaluation import eval_group_sizes
    for world in (2, 4, 8):
        for n_img, n_txt in ((1000, 1000), (1000, 5000), (29, 145)):
            gi, gt = eval_group_sizes({}, n_img, n_txt, world)
            groups_i, groups_t = -(-n_img // gi), -(-n_txt // gt)
            assert groups_i >= world and groups_t >= world, (world, n_img, n_txt, gi, gt)
            assert gi <= 256 and gt <= 1024
    assert eval_group_sizes({}, 1000, 5000, 1) == (256, 1024)                                       # one rank: the measured group sizes stay
    assert eval_group_sizes({'eval_group_images': 0, 'eval_group_texts': 0}, 1000, 1000, 8) == (0, 0)   # 0 = as the loader yields them
    assert eval_group_sizes({'eval_group_images': 64, 'eval_group_texts': 100}, 1000, 1000, 4) == (64, 100)


def test_bench_eight_rank_control_flow_dry_run_under_gloo():
    """`python bench.py --gpus 8` as the driver will start it on an 8-GPU node - bare, no launcher - but on the host: LECCR_BENCH_BACKEND=gloo
    LECCR_BENCH_DRYRUN=1.  The script starts its own 8 ranks (torch.distributed.run child, 127.0.0.1 rendezvous), every rank passes the
    world-size check, runs the barrier / timed-region / MAX-over-ranks protocol with the exchange steps' collectives on host tensors, the
    per-rank stage timers reach rank 0, and exactly ONE JSON line comes out - marked dry_run, value null (nothing about MI355X is measured
    here; round-4 verdict, next #9: the first multi-GPU lease should meet a launch path that has already run with 8 ranks)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    env.update(LECCR_BENCH_BACKEND='gloo', LECCR_BENCH_DRYRUN='1', OMP_NUM_THREADS='1')
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '8', '--steps', '3', '--warmup', '1', '--batch', '32'],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d['dry_run'] is True and d['value'] is None and d['n_gpus'] == 8 and d['config']['world'] == 8 and d['config']['global_batch'] == 256
    assert sorted(d['stage_ms_per_rank']) == [f'rank{i}' for i in range(8)]
    assert all(set(v) == {'allgather_ms', 'allreduce_exposed_ms'} for v in d['stage_ms_per_rank'].values())
    # a wrong world still refuses: 8 ranks asked, launched as 2
    env2 = dict(env, RANK='0', WORLD_SIZE='2', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT='29871')
    r2 = subprocess.Popen([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '8'], env=dict(env2), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    r3 = subprocess.Popen([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '8'], env=dict(env2, RANK='1', LOCAL_RANK='1'), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    o2, e2 = r2.communicate(timeout=300)
    r3.communicate(timeout=300)
    assert r2.returncode != 0 and o2.strip() == '' and 'world size 2 != --gpus 8' in e2
