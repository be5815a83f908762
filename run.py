#!/usr/bin/env python3
"""Counterpart of the reference's LECCR/run.py launcher: same `--task/--dist/--config/--output_dir/--bs/--seed/--epoch/
--checkpoint/--evaluate` flags (run.py:148-176); builds a one-node `torch.distributed.run` command (the reference uses
the deprecated `torch.distributed.launch --use_env`, run.py:34-79) and executes it."""
import argparse
import os
import sys


TASK_SCRIPTS = {'itr_multi30k_caption': 'image_Retrieval_caption.py', 'itr_coco_caption': 'image_Retrieval_caption.py',
                'itr_video_caption': 'video_Retrieval_caption_double_sim.py'}


def get_nproc(dist_str):
    """--dist values of run.py:34-79: 'f2' / 'f4' / 'f8' / 'l4' (first/last N GPUs), 'gpuN', '1', 'all'."""
    import torch
    n_all = max(1, torch.cuda.device_count())
    if dist_str in ('1', 'gpu0') or dist_str.startswith('gpu'):
        return 1, dist_str[3:] if dist_str.startswith('gpu') else '0'
    if dist_str == 'all':
        return n_all, None
    if dist_str[0] in 'fl' and dist_str[1:].isdigit():
        n = int(dist_str[1:])
        ids = list(range(n)) if dist_str[0] == 'f' else list(range(n_all - n, n_all))
        return n, ','.join(map(str, ids))
    raise ValueError(f'--dist {dist_str}')


def build_command(args):
    nproc, visible = get_nproc(args.dist)
    # task dispatch of run.py:134-145: itr_multi30k_caption -> image script, itr_video_caption -> video script (run_video.sh:4).
    # 'itr_coco' launches a Retrieval.py that the reference does not ship (run.py:120-131); it is refused here as there.
    if args.task not in TASK_SCRIPTS:
        raise NotImplementedError(f'task == {args.task}')
    script = TASK_SCRIPTS
    here = os.path.dirname(os.path.abspath(__file__))
    env = f'CUDA_VISIBLE_DEVICES={visible} ' if visible is not None else ''
    cmd = (f'{env}{sys.executable} -m torch.distributed.run --nnodes=1 --nproc-per-node={nproc} '
           f'--master-addr 127.0.0.1 --master-port {args.master_port} {os.path.join(here, script[args.task])} '
           f'--output_dir {args.output_dir} --bs {args.bs} --seed {args.seed} --epoch {args.epoch} '
           f'--checkpoint {args.checkpoint}')
    if args.config:
        cmd += f' --config {args.config}'
    if args.evaluate:
        cmd += ' --evaluate'
    if args.synthetic:
        cmd += ' --synthetic'
    return cmd


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument('--task', type=str, required=True)
    p.add_argument('--dist', type=str, required=True, help='see get_nproc()')
    p.add_argument('--config', default='', type=str)
    p.add_argument('--bs', default=-1, type=int, help='global batch: per GPU = bs // num_gpus')
    p.add_argument('--epoch', default=-1, type=int)
    p.add_argument('--checkpoint', default='null', type=str)
    p.add_argument('--model', default='cclm-base-ft', type=str, help='accepted for run.py:152 compatibility (default config set)')
    p.add_argument('--output_dir', default='output/itr', type=str)
    p.add_argument('--seed', default=42, type=int)
    p.add_argument('--evaluate', action='store_true')
    p.add_argument('--synthetic', action='store_true')
    p.add_argument('--master_port', default=12897, type=int)
    return p


if __name__ == '__main__':
    a = build_parser().parse_args()
    os.makedirs(a.output_dir, exist_ok=True)
    command = build_command(a)
    print(command, flush=True)
    sys.exit(os.system(command) >> 8)
