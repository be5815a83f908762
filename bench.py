#!/usr/bin/env python3
"""bench.py - LECCR train-step throughput on MI355X (image-text pairs/s), one process per GPU.

    python bench.py [--gpus N --steps K --warmup W]

N > 1: either launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` (RANK / WORLD_SIZE in
the environment), or started bare as `python bench.py --gpus N`, in which case THIS process - before it makes any GPU call -
starts the N ranks itself as a torch.distributed.run child and exits with its code.  A bare run never silently measures one
rank: the world size must equal --gpus.

A step = RetrievalModel.forward (ViT-B/16 @384 + mBERT + caption interaction + 5 losses) + backward + gradient
all-reduce + fused AdamW + LR schedule on one synthetic Multi30K-shaped batch that is already resident in HBM
(BASELINE.json configs[1]: bf16, batch 256 per GPU, 1xMI355X; weak scaling: the per-GPU batch is fixed, the contrastive
batch is all-gathered across ranks).  Prints ONE JSON line on rank 0 with the metric, a `roofline` object for the
dominant kernel family (bf16 MFMA GEMMs, timed with hipEvents on the launch stream in legs run AFTER the timed region - the timed
region itself runs with the event profiler off), an `hbm_kernels` list (achieved TB/s of the bandwidth-bound launches), a `parity`
object (the same weights on a 16-row slice against the CPU oracle: features, logits, losses, ground-truth ranks), a
`cpu_baseline` object (the CPU oracle timed on this host: BASELINE config A, bounded sample) and - default bf16 image line only - a
`reference_precision` object: the same workload re-run at the reference's fp32 precision (pairs/s, ms/step, logits error against the
oracle, ground-truth ranks, its own roofline), the one path inside north_star's 1e-3.

Other workloads (not the headline line): --model video (the reference's MSR-VTT geometry: 4096-wide frame features, F = 8)
and --model video_vit (BASELINE configs[3] as worded: 8 frames x ViT-B/16 -> vid_encoder, vw 512).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_BF16_PEAK_TFLOPS = 2500.0      # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"
MFMA_FP8_PEAK_TFLOPS = 5000.0       # dense fp8 MFMA peak (block-scaled f8f6f4 form), same table
HBM_PEAK_TBS = 8.0                  # HBM3E, same guide
F_PAIR_GFLOP = {('image', 'ViT-B/16', 'mbert'): 389.9,      # algorithmic FLOPs per pair per train step
                ('video_vit', 'ViT-B/16', 'mbert'): 2718.0,  # config D: 8 x B/16 frames + mBERT
                ('image', 'ViT-L/14', 'xlmr-large'): 1674.6}  # config E: L/14 + XLM-R-large
XLMR_LARGE = dict(vocab_size=250002, hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096,
                  max_position_embeddings=512)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=8)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--model', default='image', choices=['image', 'video', 'video_vit'])
    ap.add_argument('--batch', type=int, default=None, help='per-GPU batch (image: 256 = BASELINE configs[1]; video: 128; video_vit: 64)')
    ap.add_argument('--clip', default='ViT-B/16')
    ap.add_argument('--text', default='mbert', choices=['mbert', 'xlmr-large'],
                    help="text tower geometry: mBERT (configs A-D) | XLM-R-large shaped BERT (BASELINE configs[4]: hidden 1024, 24 layers, 16 heads, FFN 4096, vocab 250002)")
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'fp32', 'fp8'],
                    help="bf16 = BASELINE configs[1] (the headline); fp8 = BASELINE configs[4]'s fp8 GEMM operands in the encoder blocks (forward, dX and weight-gradient products)")
    ap.add_argument('--mode', default='train', choices=['train', 'eval'],
                    help="eval: the R@K half of the metric - evaluation_coarse + itm_eval (image_Retrieval_caption.py:83-163, 261-317) on a "
                         'Multi30K-test-shaped synthetic set; a "step" is one whole evaluation pass; value = images/s')
    ap.add_argument('--eval-images', type=int, default=1000, help='Multi30K / MSCOCO-1k test: 1000 images')
    ap.add_argument('--eval-texts', type=int, default=1000, help='Multi30K test (one language): 1000 texts; MSCOCO-1k: 5000')
    ap.add_argument('--eval-batch', type=int, default=6, help='batch_size_test (Retrieval_multi30k_de_ft.yaml: 6)')
    ap.add_argument('--eval-text-batch', type=int, default=32, help='batch_size_test_text (Retrieval_multi30k_de_ft.yaml: 32)')
    ap.add_argument('--eval-no-regroup', action='store_true', help='run every loader batch through the towers as it is yielded (the reference '
                    'loop); default: evaluation_coarse regroups them to 256 images / 1024 texts per forward - same scores')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-reference-precision', action='store_true', help='skip the fp32 (reference-precision) leg of the default bf16 image line')
    ap.add_argument('--ref-steps', type=int, default=10, help='timed steps of the reference-precision leg')
    ap.add_argument('--no-parity', action='store_true')
    ap.add_argument('--parity-rows', type=int, default=16, help='rows of the batch the parity probe runs through the CPU oracle (config E: 4 keeps it to minutes)')
    ap.add_argument('--roofline-steps', type=int, default=3, help='extra single-stream steps after the timed region that time every launch on its own')
    ap.add_argument('--cpu-batch', type=int, default=32, help='cpu_baseline batch (BASELINE config A: 32)')
    ap.add_argument('--cpu-steps', type=int, default=3, help='timed cpu_baseline steps after one warm-up (BASELINE.md section 3: median of 3)')
    ap.add_argument('--cpu-budget-s', type=float, default=150.0, help='cpu_baseline stops adding steps past this many seconds')
    args = ap.parse_args()
    if args.batch is None:
        args.batch = {'image': 256, 'video': 128, 'video_vit': 64}[args.model]
    return args


def maybe_self_launch(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a child `torch.distributed.run` and exit with its
    code.  Runs before this process has imported leccr_amd or touched the GPU (a process that has initialised the GPU must
    neither exec nor hand the device to children it forks)."""
    if args.gpus <= 1 or ('RANK' in os.environ and 'WORLD_SIZE' in os.environ):
        return
    import torch
    have = torch.cuda.device_count()           # counts devices without initialising the runtime on this image
    backend = os.environ.get('LECCR_BENCH_BACKEND', 'nccl')
    if backend == 'nccl' and have < args.gpus:
        print(f'bench.py: --gpus {args.gpus} but this node exposes {have} GPU(s); refusing to report a {args.gpus}-GPU number',
              file=sys.stderr)
        sys.exit(2)
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print('bench.py: starting ranks: ' + ' '.join(cmd), file=sys.stderr, flush=True)
    sys.exit(subprocess.run(cmd, env=env).returncode)


def host_cores():
    """(threads to use, description): the physical cores this process may run on - sched_getaffinity, one thread per
    (physical id, core id) pair of /proc/cpuinfo inside it, capped by a cgroup CPU quota when one is set."""
    logical = os.cpu_count() or 1
    try:
        aff = sorted(os.sched_getaffinity(0))
    except AttributeError:
        aff = list(range(logical))
    phys = set()
    try:
        cur = {}
        with open('/proc/cpuinfo') as fh:
            for line in fh.read().split('\n') + ['']:
                if ':' in line:
                    k, v = line.split(':', 1)
                    cur[k.strip()] = v.strip()
                elif cur:
                    if int(cur.get('processor', -1)) in aff:
                        phys.add((cur.get('physical id', '0'), cur.get('core id', cur.get('processor'))))
                    cur = {}
    except OSError:
        pass
    n_phys = len(phys) or len(aff)
    quota = None
    try:
        with open('/sys/fs/cgroup/cpu.max') as fh:
            q, p = fh.read().split()
            if q != 'max':
                quota = max(1, int(float(q) / float(p)))
    except (OSError, ValueError):
        pass
    n = min(n_phys, quota) if quota else n_phys
    return max(1, n), f'os.cpu_count()={logical}, affinity={len(aff)}, physical cores in affinity={n_phys}, cgroup quota={quota}'


# ------------------------------------------------------------------------------------------------ workloads
def build_workload(args, device, rank):
    """-> (model, cfg, batch on device, fwd(model, batch) -> 5 losses, description)."""
    import torch
    from leccr_amd.synthetic import MULTI30K_CFG, VATEX_CFG, make_batch, make_video_batch
    torch.manual_seed(1234)                      # identical weights on every rank (DDP broadcasts rank 0's)
    if args.model == 'image':
        from leccr_amd.models import RetrievalModel
        cfg = dict(MULTI30K_CFG)
        cfg.update(clip_name=args.clip, compute_dtype=args.dtype, clip_text_tower=False)
        if args.text == 'xlmr-large':
            cfg['text_config'] = dict(XLMR_LARGE)
        model = RetrievalModel(cfg)
        cpu = make_batch(args.batch, rank=rank, vocab=250002 if args.text == 'xlmr-large' else 119547)
        batch = {'image': cpu['image'].to(device)}
        fwd = lambda m, b: m(b['image'], b['text'], captions=b['captions'], idx=b['idx'])
        what = f'Multi30K en-de shaped, {args.clip}@384 + ' + ('mBERT' if args.text == 'mbert' else 'XLM-R-large shaped BERT')
    elif args.model == 'video':
        from leccr_amd.models.video_model_retrieval_caption import RetrievalModel
        # configs/cclm-base-ft/Retrieval_msrvtt.yaml:36-66: vision_width 4096, vision_layer 1, num_queries 2, 2+2 interaction layers
        cfg = dict(VATEX_CFG, vision_width=4096, vision_layer=1, num_queries=2, caption_ca_layer=2, compute_dtype=args.dtype)
        model = RetrievalModel(cfg)
        cpu = make_video_batch(args.batch, rank=rank, frames=8, width=4096)
        batch = {'video': cpu['video'].to(device), 'video_mask': cpu['video_mask'].to(device)}
        fwd = lambda m, b: m(b['video'], b['video_mask'], b['text'], captions=b['captions'], idx=b['idx'])
        what = 'MSR-VTT shaped (run_video.sh), 8 frames x 4096-wide features + mBERT, double-sim losses'
    else:
        from leccr_amd.models.video_frames_model import FrameViTRetrievalModel
        cfg = dict(VATEX_CFG, vision_layer=1, clip_name=args.clip, compute_dtype=args.dtype)
        model = FrameViTRetrievalModel(cfg)
        cpu = make_video_batch(args.batch, rank=rank, frames=8, width=8)
        g = torch.Generator().manual_seed(4242 + rank)
        batch = {'video': torch.randn(args.batch, 8, 3, 384, 384, generator=g).to(device), 'video_mask': cpu['video_mask'].to(device)}
        fwd = lambda m, b: m(b['video'], b['video_mask'], b['text'], captions=b['captions'], idx=b['idx'])
        what = f'VATEX shaped (BASELINE configs[3]), 8 frames x {args.clip}@384 -> vid_encoder (vw 512) + mBERT, double-sim losses'
    model.init_caption_encoder()
    batch.update(text=[t.to(device) for t in cpu['text']], captions=cpu['captions'].to(device), idx=cpu['idx'].to(device))
    return model.to(device).train(), cfg, batch, fwd, what


def train_step(model, batch, fwd, opt, sched):
    losses = fwd(model, batch)
    loss = losses[0] + losses[1] + losses[2] + losses[3] + losses[4]
    opt.zero_grad()
    loss.backward()
    opt.step()
    sched.step()
    return losses


def _oracle_state(model):
    import torch
    return {k: v.detach().cpu().float().clone() for k, v in model.state_dict().items() if not k.startswith('caption_encoder.')}


def _oracle_losses(args, cfg, sd, batch, rows):
    """The CPU oracle's five losses on rows [0, rows) of `batch` (CPU tensors) with the state dict `sd`."""
    import torch
    from oracle import leccr_oracle as O
    wl = {k: cfg[k] for k in ('weight_reg_loss', 'weight_caption_loss', 'weight_dstl_loss', 'weight_cv_loss')}
    s = slice(0, rows)
    t0, t1, c = batch['text'][0], batch['text'][1], batch['captions']
    txt = (t0.input_ids[s], t0.attention_mask[s], t1.input_ids[s], t1.attention_mask[s], c.input_ids[s], c.attention_mask[s],
           batch['idx'][s])
    if args.model == 'image':
        return O.retrieval_forward(sd, wl, batch['image'][s], *txt, text_heads=16 if args.text == 'xlmr-large' else 12)
    video = batch['video'][s]
    if args.model == 'video_vit':
        b, F = video.shape[:2]
        hidden = O.encode_image(sd, 'clip_encoder.visual.', video.reshape(b * F, *video.shape[2:]))
        video = hidden[:, 0, :].reshape(b, F, -1)
    return O.video_retrieval_forward(sd, wl, video, batch['video_mask'][s], *txt, text_heads=16 if args.text == 'xlmr-large' else 12)


def cpu_baseline(args, cfg, model):
    """The CPU oracle (oracle/leccr_oracle.py, a restatement of the reference pinned to its fixtures - kind 'port') timed on
    this host: full step = forward + backward + AdamW, same geometry, BASELINE config A's batch (32) on the physical cores
    this process may use; bounded: at most --cpu-steps timed steps after one warm-up, fewer once --cpu-budget-s is spent."""
    import torch
    from leccr_amd.synthetic import make_batch, make_video_batch
    n, how = host_cores()
    torch.set_num_threads(n)
    sd = {k: v.requires_grad_(v.is_floating_point()) for k, v in _oracle_state(model).items()}
    params = [v for v in sd.values() if v.requires_grad]
    opt = torch.optim.AdamW(params, lr=1e-5, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.01)
    b = args.cpu_batch
    if args.model == 'image':
        batch = make_batch(b, vocab=250002 if args.text == 'xlmr-large' else 119547)
    else:
        batch = make_video_batch(b, frames=8, width=4096 if args.model == 'video' else 8)
        if args.model == 'video_vit':
            batch['video'] = torch.randn(b, 8, 3, 384, 384, generator=torch.Generator().manual_seed(4242))
    times, t_begin = [], time.perf_counter()
    for it in range(1 + args.cpu_steps):
        t0 = time.perf_counter()
        losses = _oracle_losses(args, cfg, sd, batch, b)
        opt.zero_grad()
        sum(losses).backward()
        opt.step()
        times.append(time.perf_counter() - t0)
        if it >= 1 and time.perf_counter() - t_begin > args.cpu_budget_s:
            break
    timed = times[1:] or times
    t = sorted(timed)[len(timed) // 2]
    return {'value': round(b / t, 4), 'unit': 'pairs/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': f'{len(timed)} full train step(s) (fwd+bwd+AdamW) of batch {b} (BASELINE config A) after '
                      f'{1 if len(times) > 1 else 0} warm-up, same {args.clip}+{args.text} geometry, fp32; {how}',
            's_per_step': round(t, 3), 'step_times_s': [round(x, 2) for x in times]}


def _sub_batch(batch, rows):
    import torch
    sub = {k: (v[:rows] if torch.is_tensor(v) else v) for k, v in batch.items()}
    sub['text'] = [type(t)(t.input_ids[:rows], t.attention_mask[:rows]) for t in batch['text']]
    sub['captions'] = type(batch['captions'])(batch['captions'].input_ids[:rows], batch['captions'].attention_mask[:rows])
    return sub


def parity_probe(args, cfg, model, batch, fwd, rows=16, with_losses=True):
    """After the timed region: the HIP model (the benchmarked weights, eval mode = dropout off) on rows [0, rows) of the batch against
    the CPU oracle (fp32) on the same weights and inputs.  What north_star bounds is the LOGITS, so that is what is compared: the
    three L2-normalised feature matrices (image after the caption interaction, source text, target text: models/xvlm.py:241-256) and
    the three rows x rows logits matrices feat_a feat_b^T / temp of get_contrastive_loss (models/xvlm.py:273-276), max-abs error each;
    the five losses beside them (at a handful of rows and random-init towers the contrastive terms sit near ln(rows) whatever the
    encoders do - they are the weak part of the probe, the features are the strong one).  with_losses=False (W > 1): the model's
    forward gathers across ranks, so only the collective-free encoder / feature / logits part runs, on this rank's rows.
    Also the R@K half on the same rows: the integer rank of every ground-truth pair from the HIP encoders' score matrix against the
    ranks from the ORACLE encoders' score matrix (image_Retrieval_caption.py:151-152, 261-317)."""
    import numpy as np
    import torch
    from oracle import leccr_oracle as O
    was_training = model.training
    model.eval()
    rows = min(rows, batch['idx'].shape[0])
    sub = _sub_batch(batch, rows)
    feats = None
    with torch.no_grad():
        if args.model == 'image':
            cap_ids, cap_m = sub['captions'].input_ids, sub['captions'].attention_mask
            v, _ = model.get_vision_embeds(sub['image'])
            c = model.get_caption_embeds(cap_ids, cap_m)
            a_img, _, _ = model.interaction_with_caption(image_embeds=v, caption_embeds=c, key_padding_mask=~cap_m.bool())
            feats = [model.get_features(a_img.transpose(0, 1).contiguous(), None)]
            feats += [model.get_features(None, model.get_text_embeds(t.input_ids, t.attention_mask)) for t in sub['text']]
            feats = [f.float().cpu().double() for f in feats]
        got = [float(l) for l in fwd(model, sub)] if with_losses else None
    model.train(was_training)
    cpu = {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in sub.items()}
    cpu['text'] = [t.to('cpu') for t in sub['text']]
    cpu['captions'] = sub['captions'].to('cpu')
    torch.set_num_threads(host_cores()[0])
    sd = _oracle_state(model)
    heads = 16 if args.text == 'xlmr-large' else 12
    out = {'rows': rows}
    ok = True
    with torch.no_grad():
        if feats is not None:
            vis = O.encode_image(sd, 'clip_encoder.visual.', cpu['image'])
            cap = O.bert_forward(sd, 'text_encoder.', cpu['captions'].input_ids, cpu['captions'].attention_mask, heads)
            a_img, _, _ = O.interaction_with_caption(sd, vis, cap, ~cpu['captions'].attention_mask.bool())
            ref = [O.get_features(sd, image_embeds=a_img)]
            ref += [O.get_features(sd, text_embeds=O.bert_forward(sd, 'text_encoder.', t.input_ids, t.attention_mask, heads)) for t in cpu['text']]
            ref = [f.double() for f in ref]
            temp = float(sd['temp'])
            ferr = [float((a - r).abs().max()) for a, r in zip(feats, ref)]
            pairs = ((0, 1), (0, 2), (1, 2))                 # v-s, v-t, s-t: the three get_contrastive_loss calls of a step
            lerr = [float(((feats[i] @ feats[j].t()) / temp - (ref[i] @ ref[j].t()) / temp).abs().max()) for i, j in pairs]
            # fp32: north_star's bound (measured 2.5e-6 / 2.5e-5 on the split-operand path); bf16: 2 x the largest driver-run figures
            # (BENCH_r03: 2.9e-3 / 3.9e-2); fp8: builder-run figures x 2
            ftol, ltol = {'fp32': (1e-4, 1e-3), 'bf16': (6e-3, 8e-2), 'fp8': (4e-2, 0.6)}[args.dtype]
            out.update(max_abs_feature_err={'image': round(ferr[0], 7), 'text_src': round(ferr[1], 7), 'text_tgt': round(ferr[2], 7)},
                       max_abs_logits_err={'v_s': round(lerr[0], 6), 'v_t': round(lerr[1], 6), 's_t': round(lerr[2], 6)},
                       feature_tolerance=ftol, logits_tolerance=ltol, temp=round(temp, 6))
            ok = ok and max(ferr) <= ftol and max(lerr) <= ltol
            # retrieval on these rows: rank of the ground-truth pair, HIP encoders vs oracle encoders (count of strictly larger scores)
            s_hip, s_ref = (feats[0] @ feats[2].t()).numpy(), (ref[0] @ ref[2].t()).numpy()
            rk = lambda s_: np.array([int((s_[i] > s_[i, i]).sum()) for i in range(rows)])
            same = int((rk(s_hip) == rk(s_ref)).sum() + (rk(s_hip.T) == rk(s_ref.T)).sum())
            out['gt_ranks_equal_to_oracle_encoders'] = f'{same}/{2 * rows}'
        if with_losses:
            ref_l = [float(l) for l in _oracle_losses(args, cfg, sd, cpu, rows)]
            err = max(abs(a - r) for a, r in zip(got, ref_l))
            tol = {'fp32': 1e-3, 'bf16': 3e-2, 'fp8': 1e-1}[args.dtype]
            out.update(max_abs_loss_err=round(err, 6), tolerance=tol, losses_hip=[round(x, 5) for x in got],
                       losses_oracle_fp32=[round(x, 5) for x in ref_l])
            ok = ok and err <= tol
    out['ok'] = bool(ok)
    out['what'] = (f'{args.dtype} HIP path (eval mode) vs CPU oracle fp32, same weights (after the timed steps) and inputs, full '
                   f'{args.clip}@384 + {args.text} geometry; north_star bound: logits within 1e-3 in fp32')
    return out


def reference_precision_leg(args, device, rank):
    """The same workload at the REFERENCE's precision (fp32: models/xvlm.py:88 `clip_encoder.float()`; the path a reference YAML gets here, on
    the matrix cores as bf16 hi / lo operand splits - DESIGN.md section 4a): the only path inside north_star's "logits within 1e-3", so its
    throughput belongs in the driver-run line beside the bf16 headline (round-4 verdict).  Run after the bf16 model has been freed: a fresh
    model of the same seed, --ref-steps timed steps after 2 warm-ups (same barriers, profiler off), 2 single-stream steps for the GEMM family's
    rate, then the 16-row parity probe against the CPU oracle."""
    import argparse as _ap
    import ctypes as C

    import torch
    from leccr_amd import _lib, ops
    from leccr_amd.optim import create_optimizer
    from leccr_amd.parallel import attach_grad_sync
    from leccr_amd.scheduler import create_scheduler
    a32 = _ap.Namespace(**vars(args))
    a32.dtype = 'fp32'
    torch.cuda.reset_peak_memory_stats()
    model, cfg, batch, fwd, what = build_workload(a32, device, rank)
    opt = attach_grad_sync(create_optimizer(cfg['optimizer'], model)).enable_backward_fusion()
    sched = create_scheduler(dict(cfg['schedular'], step_per_epoch=1000), opt)
    sched.step()
    for _ in range(2):
        train_step(model, batch, fwd, opt, sched)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.ref_steps):
        losses = train_step(model, batch, fwd, opt, sched)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    lib = _lib.lib()
    NK = 14
    ms, work, n = (C.c_double * NK)(), (C.c_double * NK)(), (C.c_int64 * NK)()
    overlap, ops.wgrad_overlap, ops.text_overlap = (ops.wgrad_overlap, ops.text_overlap), False, False
    fusion, opt.backward_fusion = opt.backward_fusion, False
    lib.leccr_prof_enable(1)
    for _ in range(2):
        train_step(model, batch, fwd, opt, sched)
    torch.cuda.synchronize()
    lib.leccr_prof_enable(0)
    lib.leccr_prof_read_n(NK, ms, work, n)
    ops.wgrad_overlap, ops.text_overlap = overlap
    opt.backward_fusion = fusion
    gemm_ms, gemm_fl = ms[0] + ms[1], work[0] + work[1]
    issued = gemm_fl / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
    par = parity_probe(a32, cfg, model, batch, fwd, rows=args.parity_rows, with_losses=True)
    out = {'dtype': 'fp32', 'what': 'the reference\'s own precision (models/xvlm.py:88) on the matrix cores: fp32 operands as bf16 hi / lo splits, f32 accumulation, '
                                    'statistics and losses; same workload, batch and step as the headline line',
           'pairs_s': round(args.batch * args.ref_steps / dt, 2), 'ms_per_step': round(dt / args.ref_steps * 1e3, 3), 'steps': args.ref_steps, 'warmup': 2,
           'losses': [round(float(l.detach()), 4) for l in losses],
           'max_abs_logits_err': par.get('max_abs_logits_err'), 'max_abs_feature_err': par.get('max_abs_feature_err'),
           'logits_tolerance': par.get('logits_tolerance'), 'gt_ranks': par.get('gt_ranks_equal_to_oracle_encoders'),
           'max_abs_loss_err': par.get('max_abs_loss_err'), 'parity_ok': par['ok'],
           'roofline': {'bound': 'mfma', 'kernel': 'gemm256 family, fp32 operands as bf16 splits: achieved = fp32-product flops / time, peak = bf16 MFMA peak / 3',
                        'mfma_issued_tflops': round(issued, 1), 'achieved': round(issued / 3.0, 1), 'peak': round(MFMA_BF16_PEAK_TFLOPS / 3.0, 1),
                        'unit': 'TFLOP/s', 'frac': round(issued / MFMA_BF16_PEAK_TFLOPS, 4), 'gemm_ms_per_step': round(gemm_ms / 2, 3),
                        'attn_fwd_ms_per_step': round(ms[3] / 2, 3), 'attn_bwd_ms_per_step': round(ms[4] / 2, 3),
                        'split_ms_per_step': round(ms[13] / 2, 3)},
           'peak_hbm_gb': round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)}
    return out


def retrieval_probe(args, model, batch):
    """The R@K half of the metric on what this run has: the benchmarked weights (random init: recall is at chance level - there is
    no dataset and no pretrained checkpoint on this pipeline), the batch's 256 image / target-text pairs as a
    retrieval set.  What it checks is the integer ranking: itm_eval's 13 numbers from the on-device rank kernel equal the CPU
    oracle's on the same score matrix (image_Retrieval_caption.py:261-317).  (Encoder-level ranking parity - HIP encoders' scores
    against the oracle encoders' scores - is in the parity probe above on 16 rows, and at 64 x 320 in tests/test_eval_gpu.py.)"""
    import numpy as np
    import torch
    from leccr_amd import kernels as K
    from leccr_amd.evaluation import itm_eval
    from oracle import leccr_oracle as O
    was_training = model.training
    model.eval()
    with torch.no_grad():
        v, _ = model.get_vision_embeds(batch['image'])
        c = model.get_caption_embeds(batch['captions'].input_ids, batch['captions'].attention_mask)
        a_img, _, _ = model.interaction_with_caption(image_embeds=v, caption_embeds=c,
                                                     key_padding_mask=~batch['captions'].attention_mask.bool())
        fv = model.get_features(a_img.transpose(0, 1).contiguous(), None)
        ft = model.get_features(None, model.get_text_embeds(batch['text'][1].input_ids, batch['text'][1].attention_mask))
        scores = K.gemm(fv.float().contiguous(), ft.float().contiguous())          # [images, texts] = fv ft^T
    model.train(was_training)
    n = scores.shape[0]
    s_np = scores.cpu().numpy()
    t2i, i2t = {t: t for t in range(n)}, {i: [i] for i in range(n)}
    got = itm_eval(s_np, s_np.T.copy(), t2i, i2t, device=scores.device)
    ref = O.itm_eval(s_np, s_np.T.copy(), t2i, i2t)
    same = all(float(got[k]) == float(ref[k]) for k in ref)
    return {'pairs': n, 'txt_r1': got['txt_r1'], 'img_r1': got['img_r1'], 'r_mean': round(float(got['r_mean']), 4),
            'ranks_bit_exact_vs_oracle': bool(same),
            'what': 'itm_eval on the 256 x 256 image / target-text score matrix of the benchmarked (random-init) weights: recall is at '
                    'chance level by construction; the check is that the device ranking equals the oracle\'s integer ranks'}


def eval_bench(args, device, rank, world, backend, real_stdout):
    """`--mode eval`: evaluation_coarse + itm_eval (image_Retrieval_caption.py:83-163, 261-317) on a synthetic test set of the Multi30K
    test shape - `--eval-images` images with one generated caption each (L_c = 64) and `--eval-texts` texts (L_t = 32), the YAML's test
    batch sizes - with the benchmarked towers (random init).  A step = one whole pass (every text and image encoded, score matrix,
    integer ranks, the 13 recall figures); under W ranks the encoder work is rank-sharded and the embeddings gathered once
    (leccr_amd.evaluation).  Inputs are resident in HBM and pre-tokenised (the host tokenizer is outside the path).
    Checks: itm_eval's dict bit-exact against the oracle's on the same score matrix; scores of a subsample against the ORACLE's
    encoders on the same weights, with the integer ranks inside that subsample."""
    import ctypes as C
    import types

    import numpy as np
    import torch
    import torch.distributed as dist
    from leccr_amd import _lib
    from leccr_amd.evaluation import evaluation_coarse, itm_eval, eval_graphs_on
    from leccr_amd.models import RetrievalModel
    from leccr_amd.synthetic import MULTI30K_CFG, make_batch
    torch.manual_seed(1234)
    cfg = dict(MULTI30K_CFG)
    cfg.update(clip_name=args.clip, compute_dtype=args.dtype, clip_text_tower=False)
    if args.text == 'xlmr-large':
        cfg['text_config'] = dict(XLMR_LARGE)
    model = RetrievalModel(cfg)
    model.init_caption_encoder()
    model = model.to(device).eval()
    ni, nt = args.eval_images, args.eval_texts
    vocab = 250002 if args.text == 'xlmr-large' else 119547
    bi = make_batch(ni, vocab=vocab, seed=777)                                  # images + generated captions (L_c = 64)
    bt = make_batch(nt, vocab=vocab, seed=778, res=8)                            # texts (L_t = 32); its images are not used
    image = bi['image'].to(device)
    cap_ids, cap_m = bi['captions'].input_ids.to(device), bi['captions'].attention_mask.to(device)
    txt_ids, txt_m = bt['text'][1].input_ids.to(device), bt['text'][1].attention_mask.to(device)

    class Tok:                      # pre-tokenised, device-resident: 't<i>' = text i, 'c<i>' = the generated caption of image i
        def __call__(self, text, **kw):
            ids, m = (txt_ids, txt_m) if text[0][0] == 't' else (cap_ids, cap_m)
            sel = torch.tensor([int(t[1:]) for t in text], device=device)
            o = types.SimpleNamespace(input_ids=ids[sel], attention_mask=m[sel])
            o.to = lambda dev: o
            return o

    class Loader(list):
        dataset = types.SimpleNamespace(text=[f't{i}' for i in range(nt)])
    bs = args.eval_batch
    loader = Loader([(image[i:i + bs], [f'c{j}' for j in range(i, min(ni, i + bs))], None) for i in range(0, ni, bs)])
    ecfg = {'batch_size_test_text': args.eval_text_batch, 'max_tokens': 200}
    if args.eval_no_regroup:
        ecfg.update(eval_group_images=0, eval_group_texts=0)
    per = max(1, nt // ni)
    img2txt = {i: [t for t in range(i * per, min(nt, (i + 1) * per))] or [i % nt] for i in range(ni)}
    txt2img = {t: min(ni - 1, t // per) for t in range(nt)}

    def one_pass():
        i2t, t2i = evaluation_coarse(model, loader, Tok(), device, ecfg)
        return i2t, t2i, itm_eval(i2t, t2i, txt2img, img2txt, device=device)

    from leccr_amd.parallel import collective_path
    multi = collective_path(world)

    def barrier():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(1, args.warmup)):
        one_pass()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        i2t, t2i, res = one_pass()
    barrier()
    dt = time.perf_counter() - t0
    if multi:
        tt = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt)
    # forward-only roofline of the GEMM family and the attention kernels: one more pass with the event profiler on
    lib = _lib.lib()
    NK = 14
    ms, work, n = (C.c_double * NK)(), (C.c_double * NK)(), (C.c_int64 * NK)()
    barrier()
    lib.leccr_prof_enable(1)
    evaluation_coarse(model, loader, Tok(), device, dict(ecfg, eval_graphs=False))       # (launch by launch: a replayed graph gives the event profiler nothing)
    barrier()
    lib.leccr_prof_enable(0)
    lib.leccr_prof_read_n(NK, ms, work, n)
    if rank == 0:
        from oracle import leccr_oracle as O
        want = O.itm_eval(i2t, t2i, txt2img, img2txt)
        same = all(float(res[k]) == float(want[k]) for k in want)
        # a subsample through the ORACLE's encoders (CPU fp32, same weights): scores and the ranks inside the subsample
        si, st_ = min(8, ni), min(40, nt)
        torch.set_num_threads(host_cores()[0])
        sd = _oracle_state(model)
        heads = 16 if args.text == 'xlmr-large' else 12
        with torch.no_grad():
            vis = O.encode_image(sd, 'clip_encoder.visual.', bi['image'][:si])
            cap = O.bert_forward(sd, 'text_encoder.', bi['captions'].input_ids[:si], bi['captions'].attention_mask[:si], heads)
            a_img, _, _ = O.interaction_with_caption(sd, vis, cap, ~bi['captions'].attention_mask[:si].bool())
            fv = O.get_features(sd, image_embeds=a_img)
            ft = O.get_features(sd, text_embeds=O.bert_forward(sd, 'text_encoder.', bt['text'][1].input_ids[:st_], bt['text'][1].attention_mask[:st_], heads))
        ref = (fv @ ft.t()).numpy()
        sub = i2t[:si, :st_]
        serr = float(np.max(np.abs(sub - ref)))
        rk = lambda s_: np.array([[int((row > row[j]).sum()) for j in range(s_.shape[1])] for row in s_])
        ranks_same = int((rk(sub) == rk(ref)).sum())
        x3 = 3.0 if args.dtype == 'fp32' else 1.0
        gemm_ms, gemm_fl = ms[0] + ms[1], (work[0] + work[1]) / x3
        achieved = gemm_fl / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
        peak = MFMA_BF16_PEAK_TFLOPS / x3
        out = {'metric': 'evaluation images/sec (evaluation_coarse + itm_eval)', 'value': round(ni * args.steps / dt, 2), 'unit': 'images/s',
               'texts_per_s': round(nt * args.steps / dt, 2), 'n_gpus': world, 'steps': args.steps, 'warmup': max(1, args.warmup),
               'ms_per_step': round(dt / args.steps * 1e3, 2), 'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None,
               'dtype': args.dtype, 'data': 'synthetic',
               'config': {'workload': f'evaluation_coarse + itm_eval, {ni} images x {nt} texts (Multi30K-test shaped), {args.clip}@384 + {args.text}, '
                                      f'{args.dtype}, batch_size_test {bs}, batch_size_test_text {args.eval_text_batch}'
                                      + ((' (run as yielded, whole batches as replayed HIP graphs)' if eval_graphs_on(ecfg, device) else ' (run as yielded)') if args.eval_no_regroup else ' (regrouped to >= 256 images / 1024 texts per forward inside evaluation_coarse)')
                                      + ', L_t=32, L_c=64, eval mode'
                                      + (f', encoders sharded over {world} ranks' if world > 1 else ''),
                          'parallelism': f'dp{world}', 'peak_hbm_gb': round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)},
               'roofline': {'bound': 'mfma', 'kernel': 'forward GEMMs of one evaluation pass (gemm256 / 128-tile family)', 'achieved': round(achieved, 1),
                            'peak': round(peak, 1), 'unit': 'TFLOP/s', 'frac': round(achieved / peak, 4), 'traffic': None,
                            'launches_per_pass': int(n[0] + n[1]), 'gemm_ms_per_pass': round(gemm_ms, 2),
                            'attn_fwd_ms_per_pass': round(ms[3], 2),
                            'attn_fwd_tflops': round(work[3] / (ms[3] * 1e-3) / 1e12, 1) if ms[3] > 0 else None,
                            'generic_gemm_ms_per_pass': round(ms[2], 3)},
               'retrieval': {k: (float(v) if not isinstance(v, float) else v) for k, v in res.items()},
               'parity': {'ranks_bit_exact_vs_oracle': bool(same),
                          'subsample': f'{si} images x {st_} texts through the oracle encoders (CPU fp32, same weights)',
                          'max_abs_score_err': round(serr, 7), 'subsample_ranks_equal': f'{ranks_same}/{si * st_}',
                          'ok': bool(same and serr <= {'fp32': 1e-4, 'bf16': 2e-2, 'fp8': 8e-2}[args.dtype])}}
        print(json.dumps(out), file=real_stdout, flush=True)
        if not out['parity']['ok']:
            raise SystemExit(f"bench.py --mode eval: parity outside tolerance: {out['parity']}")
    if multi:
        dist.barrier()
        dist.destroy_process_group()


HBM_KINDS = ['layernorm_fwd', 'layernorm_bwd', 'colsum', 'bert_embed_fwd', 'adamw', 'dropout']


def dry_run(args, rank, world, backend, real_stdout):
    """LECCR_BENCH_DRYRUN=1 (with LECCR_BENCH_BACKEND=gloo): the N-rank CONTROL FLOW of this script without a GPU - self-launch, env://
    rendezvous, the world-size check, the barrier / timed-region / MAX-over-ranks protocol, the exchange steps' collectives on host tensors
    of the real payload sizes (packed all-gather of the loss-stage features; the gradient all-reduce in the phases of leccr_amd.parallel, on a
    1 / 64 sample of the 1.1 GB arena), the per-rank stage timers gathered to rank 0, one JSON line from rank 0 only.  It measures NOTHING about
    MI355X (`value` is null and the line says dry_run): it exists so that the first lease of an 8-GPU node meets a launch path that has run
    with 8 ranks before (tests/test_host_logic_cpu.py)."""
    import torch
    import torch.distributed as dist
    assert backend == 'gloo', 'LECCR_BENCH_DRYRUN needs LECCR_BENCH_BACKEND=gloo (it runs on the host)'
    batch = args.batch or 256
    n, vw, D = 4, 512, 256                                              # MULTI30K_CFG: queries, vision width, embed dim
    packed = torch.randn(batch, 3 * D + n * D + 2 * vw + 2)             # AllGatherPacked: 3 features, n caption queries, 2 pooled cv vectors, idx as two words
    grads = [torch.randn(int(280e6 / 64 * f)) for f in (0.7, 0.1, 0.1, 0.1)]     # early region, then the ViT region in thirds

    def barrier():
        if world > 1:
            dist.barrier()

    def step(timers=None):
        t0 = time.perf_counter()
        if world > 1:
            out = torch.empty(world * batch, packed.shape[1])
            dist.all_gather_into_tensor(out, packed)
        t1 = time.perf_counter()
        time.sleep(0.002)                                               # (stands in for the encoders)
        t2 = time.perf_counter()
        if world > 1:
            for g in grads:
                dist.all_reduce(g)
        t3 = time.perf_counter()
        if timers is not None:
            timers['allgather_ms'] = round((t1 - t0) * 1e3, 3)
            timers['allreduce_exposed_ms'] = round((t3 - t2) * 1e3, 3)
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt)
    mine = {}
    step(mine)
    stage = {f'rank{rank}': mine}
    if world > 1:
        allr = [None] * world
        dist.all_gather_object(allr, mine)
        stage = {f'rank{r}': d for r, d in enumerate(allr)}
    if rank == 0:
        out = {'metric': 'image-text pairs/sec (train step)', 'value': None, 'unit': 'pairs/s', 'n_gpus': world, 'steps': args.steps,
               'warmup': args.warmup, 'ms_per_step': None, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype,
               'data': 'synthetic', 'dry_run': True,
               'config': {'workload': 'DRY RUN on the host (gloo): control flow of the N-rank bench only, no GPU work, nothing measured',
                          'global_batch': batch * world, 'parallelism': f'dp{world}', 'world': world, 'backend': backend,
                          'host_ms_per_step_max_over_ranks': round(dt / args.steps * 1e3, 3)},
               'stage_ms_per_rank': stage}
        print(json.dumps(out), file=real_stdout, flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse()
    maybe_self_launch(args)
    # stdout carries exactly one line - the result JSON of rank 0; the reference-style "### ..." notices of create_optimizer /
    # create_scheduler and any library chatter go to stderr
    # (file descriptor 1 itself is pointed at stderr for the run: libraries print from C++ too - gloo's "[Gloo] Rank 0 is connected to ..")
    import contextlib
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), 'w')
    os.dup2(2, 1)
    with contextlib.redirect_stdout(sys.stderr):
        _main(args, real_stdout)
    real_stdout.flush()


def _main(args, real_stdout):
    import ctypes as C

    import torch
    import torch.distributed as dist
    from leccr_amd import _lib, ops
    from leccr_amd.optim import create_optimizer
    from leccr_amd.parallel import attach_grad_sync, collective_path, init_distributed_mode, rccl_cus
    from leccr_amd.scheduler import create_scheduler
    # RCCL ('nccl') over xGMI, one rank per GPU.  LECCR_BENCH_BACKEND=gloo is a rehearsal aid only: it lets two ranks share
    # the single GPU of a development box to exercise this N > 1 control flow (not a measurement configuration).
    backend = os.environ.get('LECCR_BENCH_BACKEND', 'nccl')
    rank, world, local = init_distributed_mode(backend)
    if world != args.gpus:
        raise SystemExit(f'bench.py: world size {world} != --gpus {args.gpus} (launch with torch.distributed.run '
                         f'--nproc-per-node {args.gpus}, or run `python bench.py --gpus {args.gpus}` bare and let it start the ranks)')
    multi = collective_path(world)      # world > 1, or LECCR_ONE_RANK_WORLD=1: a one-rank RCCL world that takes the collective path (rehearsal)
    if os.environ.get('LECCR_BENCH_DRYRUN') == '1':
        return dry_run(args, rank, world, backend, real_stdout)
    if backend != 'nccl':
        local = local % max(1, torch.cuda.device_count())
    device = torch.device('cuda', local)
    torch.cuda.set_device(device)
    if args.mode == 'eval':
        return eval_bench(args, device, rank, world, backend, real_stdout)
    model, cfg, batch, fwd, what = build_workload(args, device, rank)
    opt = attach_grad_sync(create_optimizer(cfg['optimizer'], model))
    if os.environ.get('LECCR_BENCH_NO_FUSION') != '1':              # (A/B: the early region's AdamW update beside the ViT backward)
        opt.enable_backward_fusion()
    sched = create_scheduler(dict(cfg['schedular'], step_per_epoch=1000), opt)
    sched.step()                                                  # leave lr = 0 (first LambdaLR value) behind

    def barrier():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        train_step(model, batch, fwd, opt, sched)
    lib = _lib.lib()
    # The timed region runs with the in-library event profiler OFF (round 2 timed it with two hipEvents around each of ~700 launches
    # per step); K + 1 stream events mark the step boundaries for the median (no host sync inside the region).
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    barrier()
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        losses = train_step(model, batch, fwd, opt, sched)
        marks[i + 1].record()
    barrier()
    dt = time.perf_counter() - t0
    step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    median_ms = step_ms[len(step_ms) // 2] if len(step_ms) % 2 else 0.5 * (step_ms[len(step_ms) // 2 - 1] + step_ms[len(step_ms) // 2])
    if multi:
        tt = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt)
    NK = 14
    ms, work, n = (C.c_double * NK)(), (C.c_double * NK)(), (C.c_int64 * NK)()
    busy, gwork, gn = C.c_double(), C.c_double(), C.c_int64()
    prof_steps = args.roofline_steps if args.roofline_steps > 0 else 0
    prof_ms_per_step, dropped = None, 0
    if prof_steps:
        # Profiled leg, SAME schedule as the timed region (side streams, fused update), event profiler ON: the union of the GEMM launch
        # intervals over all streams (weight-gradient GEMMs run beside the dX chain: summing durations would count the overlapped
        # stretches twice) and the cost of the profiler itself, as the A/B against the unprofiled timed region.
        barrier()
        lib.leccr_prof_enable(1)
        tp = time.perf_counter()
        for _ in range(prof_steps):
            train_step(model, batch, fwd, opt, sched)
        barrier()
        prof_ms_per_step = (time.perf_counter() - tp) * 1e3 / prof_steps
        lib.leccr_prof_enable(0)
        dropped += int(lib.leccr_prof_dropped())
        lib.leccr_prof_read_n(NK, ms, work, n)
        lib.leccr_prof_read_union(0b11 | (0b11 << 11), C.byref(busy), C.byref(gwork), C.byref(gn))
    # Roofline leg: the same steps with every launch on ONE stream, so that a launch's event interval is the kernel running
    # alone (in the timed region above a dW GEMM shares the chip with the dX chain, and its own interval is stretched by that).
    ms1, work1, n1 = (C.c_double * NK)(), (C.c_double * NK)(), (C.c_int64 * NK)()
    if args.roofline_steps > 0:
        overlap, ops.wgrad_overlap, ops.text_overlap = (ops.wgrad_overlap, ops.text_overlap), False, False
        fusion, opt.backward_fusion = opt.backward_fusion, False
        barrier()
        lib.leccr_prof_enable(1)
        for _ in range(args.roofline_steps):
            train_step(model, batch, fwd, opt, sched)
        barrier()
        lib.leccr_prof_enable(0)
        dropped += int(lib.leccr_prof_dropped())
        lib.leccr_prof_read_n(NK, ms1, work1, n1)
        ops.wgrad_overlap, ops.text_overlap = overlap
        opt.backward_fusion = fusion
        leg, leg_ms, leg_work, leg_n = args.roofline_steps, ms1, work1, n1
    else:
        leg, leg_ms, leg_work, leg_n = 1, ms1, work1, n1              # (no profiled leg at all: the roofline object reports zeros)
    # Stage timers of the exchange steps (leccr_amd.diag): two more steps with stream events around the packed all-gather, the loss
    # stage, the exposed part of the gradient all-reduce and the backward pass - every rank's figures go into the JSON, so that the first
    # multi-GPU run yields a diagnosis (which rank waits where) and not only a curve.  After the timed region; never inside it.
    stage_ms = None
    if multi or os.environ.get('LECCR_BENCH_DIAG') == '1':
        from leccr_amd import diag
        diag.enabled = True
        diag.reset()
        barrier()
        for _ in range(2):
            losses_d = fwd(model, batch)
            loss_d = losses_d[0] + losses_d[1] + losses_d[2] + losses_d[3] + losses_d[4]
            opt.zero_grad()
            diag.begin('backward')
            loss_d.backward()
            diag.end('backward')
            diag.begin('optimizer_step')
            opt.step()
            diag.end('optimizer_step')
            sched.step()
        mine = diag.report()
        diag.enabled = False
        stage_ms = {f'rank{rank}': mine}
        if multi:
            try:                                    # (diagnostics must never cost the run its result line)
                allr = [None] * world
                dist.all_gather_object(allr, mine)
                stage_ms = {f'rank{r}': d for r, d in enumerate(allr)}
            except Exception as e:                  # noqa: BLE001
                print(f'bench.py: stage timers of the other ranks unavailable ({e})', file=sys.stderr)
    parity = None
    retrieval = None
    ref_prec = None
    if not args.no_parity and (world == 1 or (rank == 0 and args.model == 'image')):
        # W > 1: the model's forward gathers across ranks, so rank 0 runs only the collective-free part on its own rows (encoders ->
        # features -> logits against the oracle); the five losses are compared on one rank only
        parity = parity_probe(args, cfg, model, batch, fwd, rows=args.parity_rows, with_losses=world == 1)
        if args.model == 'image' and world == 1:
            retrieval = retrieval_probe(args, model, batch)
    if rank == 0:
        G = args.batch * world
        value = G * args.steps / dt
        union_ms, union_fl = busy.value, gwork.value
        # the GEMM family: bf16 launches (kinds 0, 1) and, with --dtype fp8, the fp8-operand launches (kinds 11, 12).  A mixed step is
        # priced launch by launch: `peak` is the rate at which the SAME mix of flops would run with every launch at its own MFMA peak
        # (fp8 flops at 5 PF, bf16 flops at 2.5 PF), so frac = (time at peak) / (time measured)
        bf_ms, bf_fl, bf_n = leg_ms[0] + leg_ms[1], leg_work[0] + leg_work[1], leg_n[0] + leg_n[1]
        f8_ms, f8_fl, f8_n = leg_ms[11] + leg_ms[12], leg_work[11] + leg_work[12], leg_n[11] + leg_n[12]
        gemm_ms, gemm_fl, gemm_n = bf_ms + f8_ms, bf_fl + f8_fl, bf_n + f8_n
        # --dtype fp32: every large product is ONE bf16 launch over hi / lo splits of its f32 operands (3 x the reduction length:
        # kernels.f32_mfma, leccr_split3).  The launches' own flop counts are the bf16 flops ISSUED; the algorithmic flops of the fp32
        # products are a third of that, and the rate this formulation can reach at the bf16 MFMA peak is a third of 2.5 PF
        x3 = 3.0 if args.dtype == 'fp32' else 1.0
        issued = gemm_fl / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
        gemm_fl, bf_fl = gemm_fl / x3, bf_fl / x3
        achieved = issued / x3
        t_peak = bf_fl / (MFMA_BF16_PEAK_TFLOPS / x3) + f8_fl / MFMA_FP8_PEAK_TFLOPS
        peak = gemm_fl / t_peak if t_peak > 0 else MFMA_BF16_PEAK_TFLOPS
        mixed = None
        if f8_n:
            mixed = {'fp8': {'launches_per_step': f8_n // max(1, leg), 'ms_per_step': round(f8_ms / max(1, leg), 3),
                             'tflops': round(f8_fl / (f8_ms * 1e-3) / 1e12, 1) if f8_ms > 0 else None, 'peak': MFMA_FP8_PEAK_TFLOPS,
                             'weight_gradient_tflops': round(leg_work[12] / (leg_ms[12] * 1e-3) / 1e12, 1) if leg_ms[12] > 0 else None},
                     'bf16': {'launches_per_step': bf_n // max(1, leg), 'ms_per_step': round(bf_ms / max(1, leg), 3),
                              'tflops': round(bf_fl / (bf_ms * 1e-3) / 1e12, 1) if bf_ms > 0 else None, 'peak': MFMA_BF16_PEAK_TFLOPS}}
        traffic, traffic_src = None, None
        suf = '' if args.dtype == 'bf16' else '_' + args.dtype
        tj = next((p for p in (os.path.join(ROOT, 'profiles', f'r{r:02d}_gemm256_hbm_traffic{suf}.json') for r in (5, 4, 3, 2, 1))
                   if os.path.exists(p)), '')
        if tj and args.model == 'image' and args.clip == 'ViT-B/16' and args.batch == 256 and args.dtype in ('bf16', 'fp32'):
            with open(tj) as fh:        # PMC counters cannot be read from inside the run: the committed rocprofv3 passes
                tr = json.load(fh)      # of this same command (FETCH_SIZE x2 + WRITE_SIZE, per launch) are reported
            traffic, traffic_src = round(tr['hbm_bytes_per_launch']), 'profiles/' + os.path.basename(tj)
        hbm = []
        for k, name in list(enumerate(HBM_KINDS, start=5)) + [(13, 'split3 (f32 -> bf16 hi / lo copies)')]:
            if leg_n[k] and leg_ms[k] > 0:
                tbs = leg_work[k] / (leg_ms[k] * 1e-3) / 1e12
                hbm.append({'kernel': name, 'launches_per_step': leg_n[k] // leg, 'ms_per_step': round(leg_ms[k] / leg, 3),
                            'algorithmic_gb_per_step': round(leg_work[k] / leg / 1e9, 3), 'achieved_tb_s': round(tbs, 3),
                            'frac_of_hbm_peak': round(tbs / HBM_PEAK_TBS, 3)})
        try:
            rccl = '.'.join(map(str, torch.cuda.nccl.version())) if multi else None
        except Exception:                                                            # noqa: BLE001 (reporting only)
            rccl = None
        fpair = F_PAIR_GFLOP.get((args.model, args.clip, args.text))
        out = {
            'metric': 'image-text pairs/sec (train step)' if args.model == 'image' else 'video-text pairs/sec (train step)',
            'value': round(value, 2), 'unit': 'pairs/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 3),
            'ms_per_step_median': round(median_ms, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': f'{what}, {args.dtype}, batch {args.batch}/GPU (global {G}), L_t=32, L_c=64, train mode '
                                   f'(dropout 0.1 on), full step (fwd+bwd+all-gather+all-reduce+AdamW)',
                       'global_batch': G, 'parallelism': f'dp{world}',
                       'step_tflops_algorithmic': round(fpair * value / 1e3, 1) if fpair else None,
                       'losses': [round(float(l.detach()), 4) for l in losses],
                       'peak_hbm_gb': round(torch.cuda.max_memory_allocated() / 2 ** 30, 1),
                       'peak_hbm_reserved_gb': round(torch.cuda.max_memory_reserved() / 2 ** 30, 1),
                       'alloc_retries': int(torch.cuda.memory_stats().get('num_alloc_retries', 0)),
                       'world': world, 'backend': backend if multi else None, 'rccl_version': rccl,
                       'NCCL_MAX_NCHANNELS': os.environ.get('NCCL_MAX_NCHANNELS'), 'LECCR_RCCL_CUS': rccl_cus()},
            'roofline': {'bound': 'mfma', 'kernel': ('gemm256 family (256x256 MFMA GEMMs: 8-wave NT forward / dX, 4-wave TN weight gradient'
                                                     + (', fp8-operand 4-wave NT / TN' if args.dtype == 'fp8' else '') + ')'
                                                     + ('; fp32 operands as bf16 hi / lo splits: achieved = fp32-product flops / time, peak = bf16 MFMA peak / 3 '
                                                        '(three bf16 products per fp32 product; the f32 MFMA instruction peaks at 157)' if args.dtype == 'fp32' else '')),
                         'mfma_issued_tflops': round(issued, 1),
                         'achieved': round(achieved, 1), 'peak': round(peak, 1), 'unit': 'TFLOP/s', 'by_operand_type': mixed,
                         'frac': round(achieved / peak, 4), 'traffic': traffic,
                         'traffic_unit': 'HBM bytes per launch (PMC)', 'traffic_source': traffic_src,
                         'launches_per_step': gemm_n // max(1, leg),
                         'avg_launch_us': round(gemm_ms * 1e3 / max(1, gemm_n), 2),
                         'avg_launch_gflop': round(gemm_fl / max(1, gemm_n) / 1e9, 2),
                         'timing': (f'HIP events on the launch stream around every GEMM launch of {leg} single-stream steps run right '
                                    'after the timed region (same process, same tensors): each interval is the kernel alone. In the timed '
                                    'region the dW launches and the text towers share the chip with the image tower / dX chain on side streams')
                                   if args.roofline_steps > 0 else 'HIP events around every GEMM launch of the timed region',
                         'overlapped_gemm_tflops_union': round(union_fl / (union_ms * 1e-3) / 1e12, 1) if union_ms > 0 else None,
                         'overlapped_gemm_busy_share': round(union_ms / (prof_ms_per_step * prof_steps), 3) if prof_steps else None,
                         'profiler': {'timed_region': 'off', 'ms_per_step_profiler_off': round(dt / args.steps * 1e3, 3),
                                      'ms_per_step_profiler_on': round(prof_ms_per_step, 3) if prof_steps else None,
                                      'records_dropped': dropped},
                         'gemm_ms_per_step': round(gemm_ms / leg, 3),
                         'attn_fwd_ms_per_step': round(leg_ms[3] / leg, 3), 'attn_bwd_ms_per_step': round(leg_ms[4] / leg, 3),
                         'attn_fwd_tflops': round(leg_work[3] / (leg_ms[3] * 1e-3) / 1e12, 1) if leg_ms[3] > 0 else None,
                         'attn_bwd_tflops': round(leg_work[4] / (leg_ms[4] * 1e-3) / 1e12, 1) if leg_ms[4] > 0 else None,
                         'generic_gemm_ms_per_step': round(leg_ms[2] / leg, 3)},
            'hbm_kernels': hbm,
        }
        if stage_ms is not None:
            out['stage_ms_per_rank'] = stage_ms
        if parity is not None:
            out['parity'] = parity
        if retrieval is not None:
            out['retrieval'] = retrieval
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(args, cfg, model)
        if world == 1 and args.model == 'image' and args.dtype == 'bf16' and not args.no_reference_precision:
            # the reference-precision line of the same workload (after everything that needs the bf16 model: its memory is handed back first)
            del model, opt, sched, batch, fwd, losses
            import gc
            gc.collect()
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
            ref_prec = reference_precision_leg(args, device, rank)
            out['reference_precision'] = ref_prec
        print(json.dumps(out), file=real_stdout, flush=True)
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    if parity is not None and not parity['ok']:
        raise SystemExit(f'bench.py: parity probe outside its tolerance: {parity}')
    if retrieval is not None and not retrieval['ranks_bit_exact_vs_oracle']:
        raise SystemExit(f'bench.py: device ranking differs from the oracle: {retrieval}')
    if ref_prec is not None and not ref_prec['parity_ok']:
        raise SystemExit(f'bench.py: reference-precision parity probe outside its tolerance: {ref_prec}')


if __name__ == '__main__':
    main()
