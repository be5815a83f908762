#!/usr/bin/env python3
"""bench.py - LECCR train-step throughput on MI355X (image-text pairs/s), one process per GPU.

    python bench.py [--gpus N --steps K --warmup W]        (N>1: launched by torch.distributed.run, one rank per GPU)

A step = RetrievalModel.forward (ViT-B/16 @384 + mBERT + caption interaction + 5 losses) + backward + gradient
all-reduce + fused AdamW + LR schedule on one synthetic Multi30K-shaped batch that is already resident in HBM
(BASELINE.json configs[1]: bf16, batch 256 per GPU, 1xMI355X; weak scaling: the per-GPU batch is fixed, the contrastive
batch is all-gathered across ranks).  Prints ONE JSON line on rank 0 with the metric, a `roofline` object for the
dominant kernel family (bf16 MFMA GEMMs, timed with hipEvents on the launch stream inside the timed region) and a
`cpu_baseline` object (the CPU oracle timed on this host, bounded sample).
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_BF16_PEAK_TFLOPS = 2500.0      # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"
F_PAIR_GFLOP = 389.9                # algorithmic FLOPs per pair per train step, B/16 + mBERT


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=8)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=256, help='per-GPU batch (BASELINE configs[1]: 256)')
    ap.add_argument('--clip', default='ViT-B/16')
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'fp32'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--roofline-steps', type=int, default=3, help='extra single-stream steps after the timed region that time the GEMM launches on their own')
    ap.add_argument('--cpu-batch', type=int, default=4)
    ap.add_argument('--cpu-steps', type=int, default=6)
    return ap.parse_args()


def build_model(args, device):
    from leccr_amd.models import RetrievalModel
    from leccr_amd.synthetic import MULTI30K_CFG
    cfg = dict(MULTI30K_CFG)
    cfg.update(clip_name=args.clip, compute_dtype=args.dtype, clip_text_tower=False)
    torch.manual_seed(1234)                      # identical weights on every rank (DDP broadcasts rank 0's)
    model = RetrievalModel(cfg)
    model.init_caption_encoder()
    return model.to(device).train(), cfg


def train_step(model, batch, opt, sched):
    losses = model(batch['image'], batch['text'], captions=batch['captions'], idx=batch['idx'])
    loss = losses[0] + losses[1] + losses[2] + losses[3] + losses[4]
    opt.zero_grad()
    loss.backward()
    opt.step()
    sched.step()
    return losses


def cpu_baseline(args, cfg):
    """The CPU oracle (oracle/leccr_oracle.py, a restatement of the reference pinned to its fixtures - kind 'port')
    timed on this host: full step = forward + backward + AdamW, same geometry, batch `cpu_batch`."""
    from leccr_amd.models import RetrievalModel
    from leccr_amd.synthetic import make_batch
    from oracle import leccr_oracle as O
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))      # the GPU box's CPU share for one GPU
    torch.manual_seed(1234)
    c = dict(cfg)
    c['compute_dtype'] = 'fp32'
    model = RetrievalModel(c)
    model.init_caption_encoder()
    sd = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in model.state_dict().items()
          if not k.startswith('caption_encoder.')}
    del model
    params = [v for v in sd.values() if v.requires_grad]
    opt = torch.optim.AdamW(params, lr=1e-5, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.01)
    b = args.cpu_batch
    batch = make_batch(b)
    wl = {k: cfg[k] for k in ('weight_reg_loss', 'weight_caption_loss', 'weight_dstl_loss', 'weight_cv_loss')}
    times = []
    for it in range(1 + args.cpu_steps):
        t0 = time.perf_counter()
        losses = O.retrieval_forward(sd, wl, batch['image'], batch['text'][0].input_ids, batch['text'][0].attention_mask,
                                     batch['text'][1].input_ids, batch['text'][1].attention_mask,
                                     batch['captions'].input_ids, batch['captions'].attention_mask, batch['idx'],
                                     text_heads=12)
        opt.zero_grad()
        sum(losses).backward()
        opt.step()
        times.append(time.perf_counter() - t0)
    t = sorted(times[1:])[len(times[1:]) // 2]
    return {'value': round(b / t, 4), 'unit': 'pairs/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': f'{args.cpu_steps} full train steps (fwd+bwd+AdamW) of batch {b} after 1 warm-up, same '
                      f'{args.clip}+mBERT geometry, fp32, os.cpu_count()={os.cpu_count()}', 's_per_step': round(t, 3)}


def main():
    # stdout carries exactly one line - the result JSON of rank 0; the reference-style "### ..." notices of create_optimizer /
    # create_scheduler and any library chatter go to stderr
    import contextlib
    real_stdout = sys.stdout
    with contextlib.redirect_stdout(sys.stderr):
        _main(real_stdout)


def _main(real_stdout):
    args = parse()
    from leccr_amd import _lib
    from leccr_amd.optim import create_optimizer
    from leccr_amd.parallel import attach_grad_sync, init_distributed_mode
    from leccr_amd.scheduler import create_scheduler
    from leccr_amd.synthetic import make_batch
    import ctypes as C
    # RCCL ('nccl') over xGMI, one rank per GPU.  LECCR_BENCH_BACKEND=gloo is a rehearsal aid only: it lets two ranks share
    # the single GPU of a development box to exercise this N > 1 control flow (not a measurement configuration).
    backend = os.environ.get('LECCR_BENCH_BACKEND', 'nccl')
    rank, world, local = init_distributed_mode(backend)
    assert world == args.gpus or world == 1, (world, args.gpus)
    if backend != 'nccl':
        local = local % max(1, torch.cuda.device_count())
    device = torch.device('cuda', local)
    torch.cuda.set_device(device)
    model, cfg = build_model(args, device)
    opt = attach_grad_sync(create_optimizer(cfg['optimizer'], model)).enable_backward_fusion()
    sched = create_scheduler(dict(cfg['schedular'], step_per_epoch=1000), opt)
    sched.step()                                                  # leave lr = 0 (first LambdaLR value) behind
    cpu = make_batch(args.batch, rank=rank)
    batch = {'image': cpu['image'].to(device), 'text': [t.to(device) for t in cpu['text']],
             'captions': cpu['captions'].to(device), 'idx': cpu['idx'].to(device)}

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        train_step(model, batch, opt, sched)
    lib = _lib.lib()
    barrier()
    lib.leccr_prof_enable(1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        losses = train_step(model, batch, opt, sched)
    barrier()
    dt = time.perf_counter() - t0
    lib.leccr_prof_enable(0)
    if world > 1:
        tt = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt)
    ms, work, n = (C.c_double * 5)(), (C.c_double * 5)(), (C.c_int64 * 5)()
    lib.leccr_prof_read(ms, work, n)
    # GEMM busy time = union of the launch intervals (HIP events on the launch streams): the weight-gradient GEMMs run on a
    # second stream beside the dX chain, so summing per-launch durations would count the overlapped stretches twice
    busy, gwork, gn = C.c_double(), C.c_double(), C.c_int64()
    lib.leccr_prof_read_union(0b11, C.byref(busy), C.byref(gwork), C.byref(gn))
    # Roofline leg: the same steps with every launch on ONE stream, so that a GEMM launch's event interval is the kernel running
    # alone (in the timed region above a dW GEMM shares the chip with the dX chain, and its own interval is stretched by that).
    from leccr_amd import ops
    ms1, work1, n1 = (C.c_double * 5)(), (C.c_double * 5)(), (C.c_int64 * 5)()
    if args.roofline_steps > 0:
        overlap, ops.wgrad_overlap, ops.text_overlap = (ops.wgrad_overlap, ops.text_overlap), False, False
        fusion, opt.backward_fusion = opt.backward_fusion, False
        barrier()
        lib.leccr_prof_enable(1)
        for _ in range(args.roofline_steps):
            train_step(model, batch, opt, sched)
        barrier()
        lib.leccr_prof_enable(0)
        lib.leccr_prof_read(ms1, work1, n1)
        ops.wgrad_overlap, ops.text_overlap = overlap
        opt.backward_fusion = fusion
    if rank == 0:
        G = args.batch * world
        value = G * args.steps / dt
        union_ms, union_fl = busy.value, gwork.value
        if args.roofline_steps > 0:
            gemm_ms, gemm_fl, gemm_n, leg_steps = ms1[0] + ms1[1], work1[0] + work1[1], n1[0] + n1[1], args.roofline_steps
        else:
            gemm_ms, gemm_fl, gemm_n, leg_steps = ms[0] + ms[1], work[0] + work[1], n[0] + n[1], args.steps
        achieved = gemm_fl / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
        traffic, traffic_src = None, None
        tj = os.path.join(ROOT, 'profiles', 'r01_gemm256_hbm_traffic.json')
        if os.path.exists(tj) and args.clip == 'ViT-B/16' and args.batch == 256 and args.dtype == 'bf16':
            with open(tj) as fh:        # PMC counters cannot be read from inside the run: the committed rocprofv3 passes
                tr = json.load(fh)      # of this same command (FETCH_SIZE x2 + WRITE_SIZE, per launch) are reported
            traffic, traffic_src = round(tr['hbm_bytes_per_launch']), 'profiles/r01_gemm256_hbm_traffic.json'
        out = {
            'metric': 'image-text pairs/sec (train step)', 'value': round(value, 2), 'unit': 'pairs/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': f'Multi30K en-de shaped, {args.clip}@384 + mBERT, {args.dtype}, batch {args.batch}/GPU '
                                   f'(global {G}), L_t=32, L_c=64, train mode (dropout 0.1 on), full step (fwd+bwd+all-gather+all-reduce+AdamW)',
                       'global_batch': G, 'parallelism': f'dp{world}',
                       'step_tflops_algorithmic': round(F_PAIR_GFLOP * value / 1e3, 1) if args.clip == 'ViT-B/16' else None,
                       'losses': [round(float(l.detach()), 4) for l in losses],
                       'peak_hbm_gb': round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)},
            'roofline': {'bound': 'mfma', 'kernel': 'gemm256_kernel (bf16 256x256x64 MFMA GEMM: NT forward/dX + TN weight-gradient instantiations)',
                         'achieved': round(achieved, 1), 'peak': MFMA_BF16_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': round(achieved / MFMA_BF16_PEAK_TFLOPS, 4), 'traffic': traffic,
                         'traffic_unit': 'HBM bytes per launch (PMC)', 'traffic_source': traffic_src,
                         'launches_per_step': gemm_n // max(1, leg_steps),
                         'avg_launch_us': round(gemm_ms * 1e3 / max(1, gemm_n), 2),
                         'avg_launch_gflop': round(gemm_fl / max(1, gemm_n) / 1e9, 2),
                         'timing': (f'HIP events on the launch stream around every GEMM launch of {leg_steps} single-stream steps run right '
                                    'after the timed region (same process, same tensors): each interval is the kernel alone. In the timed '
                                    'region the dW launches and the text towers share the chip with the image tower / dX chain on side streams')
                                   if args.roofline_steps > 0 else 'HIP events around every GEMM launch of the timed region',
                         'timed_region_gemm_tflops_union': round(union_fl / (union_ms * 1e-3) / 1e12, 1) if union_ms > 0 else None,
                         'timed_region_gemm_busy_share': round(union_ms / (dt * 1e3), 3),
                         'attn_fwd_ms_per_step': round((ms1[3] / leg_steps) if args.roofline_steps > 0 else (ms[3] / args.steps), 3),
                         'attn_bwd_ms_per_step': round((ms1[4] / leg_steps) if args.roofline_steps > 0 else (ms[4] / args.steps), 3),
                         'generic_gemm_ms_per_step': round((ms1[2] / leg_steps) if args.roofline_steps > 0 else (ms[2] / args.steps), 3)},
        }
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(args, cfg)
        print(json.dumps(out), file=real_stdout, flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
