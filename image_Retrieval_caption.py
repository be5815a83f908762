#!/usr/bin/env python3
"""Counterpart of the reference's LECCR/image_Retrieval_caption.py (same CLI, same train / evaluation_coarse /
itm_eval / main structure) running on leccr_amd.

    python3 -m torch.distributed.run --nproc_per_node=2 image_Retrieval_caption.py --config cfg.yaml --bs 128 ...
    python3 image_Retrieval_caption.py --synthetic --bs 32 --epoch 1        # no dataset needed (seeded batches)

Real data: the reference's `dataset` package (create_dataset / create_loader / mbert_build_tokenizer) must be importable
(it is outside this repo's scope: JPEG decoding + HF tokenizer).  `--synthetic` replaces loader and tokenizer by seeded
Multi30K-shaped batches (leccr_amd.synthetic) and is what the tests exercise.
"""
import argparse
import datetime
import json
import os
import time
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

from leccr_amd import utils
from leccr_amd.evaluation import evaluation_coarse, itm_eval          # noqa: F401  (same names as the reference)
from leccr_amd.models.model_retrieval_caption import RetrievalModel
from leccr_amd.optim import create_optimizer
from leccr_amd.parallel import attach_grad_sync
from leccr_amd.pipeline import DeferredScalars, DevicePrefetcher, TokenizingCollate, tokenize_or_pass
from leccr_amd.scheduler import create_scheduler
from leccr_amd.synthetic import MULTI30K_CFG, make_batch


class _PassThroughTokenizer:
    """--synthetic: 'texts' are already Encoding objects."""

    def __call__(self, text, **kw):
        return text


def clip_tokenize(texts):
    """caption_encoder_name 'clip': the reference tokenises generated captions with its own `clip` package (clip.tokenize,
    image_Retrieval_caption.py:26, 56, 129) - like the `dataset` package it must be importable (the BPE vocabulary ships with it)."""
    from clip import tokenize
    return tokenize(texts, truncate=True)


def train(model, data_loader, optimizer, tokenizer, epoch, device, scheduler, config):
    """image_Retrieval_caption.py:28-81."""
    model.train()
    metric_logger = utils.MetricLogger(delimiter='  ')
    metric_logger.add_meter('lr', utils.SmoothedValue(window_size=1, fmt='{value:.6f}'))
    for name in ('loss_itc_vs', 'loss_itc_vt', 'loss_itc_st', 'loss_itc_c', 'loss_reg_c'):
        metric_logger.add_meter(name, utils.SmoothedValue(window_size=1, fmt='{value:.4f}'))
    header = 'Train Epoch: [{}]'.format(epoch)
    names = ('loss_itc_vs', 'loss_itc_vt', 'loss_itc_st', 'loss_itc_c', 'loss_reg_c')
    deferred = DeferredScalars()          # loss scalars reach the logger one step late, without a host sync per step
    batches = DevicePrefetcher(data_loader, device)       # batch k+1 is pinned + copied on a side stream during step k
    for i, (image, text_ls, captions, idx, cap_idx) in enumerate(metric_logger.log_every(batches, 50, header)):
        # tokenised in the loader workers (--tokenize_in_workers: TokenizingCollate) or here, as the reference does
        text_inputs = [tokenize_or_pass(tokenizer, text, config['max_tokens'], device) for text in text_ls]
        if config.get('caption_encoder_name', 'mbert') == 'clip':      # image_Retrieval_caption.py:55-56: CLIP token ids [b, 77]
            captions = captions.to(device) if torch.is_tensor(captions) else clip_tokenize(captions).to(device)
        else:
            captions = tokenize_or_pass(tokenizer, captions, config['max_tokens'], device)
        losses = model(image, text_inputs, captions=captions, idx=idx, epoch=epoch, cap_idx=cap_idx)
        loss = losses[0] + losses[1] + losses[2] + losses[3] + losses[4]
        optimizer.zero_grad()
        loss.backward()
        optimizer.step()
        scheduler.step()
        vals = deferred.push(list(losses))
        if vals is not None:
            metric_logger.update(lr=optimizer.param_groups[0]['lr'], **dict(zip(names, vals)))
    vals = deferred.flush()
    if vals is not None:
        metric_logger.update(lr=optimizer.param_groups[0]['lr'], **dict(zip(names, vals)))
    metric_logger.synchronize_between_processes()
    print('Averaged stats:', metric_logger.global_avg())
    return {k: '{:.5f}'.format(meter.global_avg) for k, meter in metric_logger.meters.items()}


class _SyntheticTrain:
    def __init__(self, steps, b, rank, geometry):
        self.steps, self.b, self.rank, self.g = steps, b, rank, geometry

    def __len__(self):
        return self.steps

    def __iter__(self):
        for s in range(self.steps):
            d = make_batch(self.b, rank=self.rank, seed=42 + 1000 * s, **self.g)
            yield d['image'], d['text'], d['captions'], d['idx'], None


def main(args, config):
    utils.init_distributed_mode(args)
    device = torch.device(args.device)
    world_size = utils.get_world_size()
    if args.epoch > 0:
        config['schedular']['epochs'] = args.epoch
        print(f'### set epochs to: {args.epoch}', flush=True)
    if args.bs > 0:
        config['batch_size_train'] = args.bs // world_size
    seed = args.seed + utils.get_rank()
    torch.manual_seed(seed)
    np.random.seed(seed)
    config['seed'] = args.seed

    if args.synthetic:
        geo = dict(res=config.get('clip_image_res', 384), L_t=config.get('synthetic_L_t', 32),
                   L_c=config.get('synthetic_L_c', 64), vocab=config.get('synthetic_vocab', 119547))
        train_loader = _SyntheticTrain(args.synthetic_steps, config['batch_size_train'], utils.get_rank(), geo)
        tokenizer = _PassThroughTokenizer()
        test_loaders = {}
    else:
        from dataset import create_dataset, create_sampler, create_loader, mbert_build_tokenizer   # reference package
        train_dataset, val_dataset_dict, test_dataset_dict = create_dataset('re_caption', config, args.evaluate)
        samplers = create_sampler([train_dataset], [True], world_size, utils.get_rank()) + [None]
        tokenizer = mbert_build_tokenizer()
        # --tokenize_in_workers: the three tokenizer calls of a step run in the loader's worker processes (pipeline.TokenizingCollate)
        train_collate = TokenizingCollate(tokenizer, config['max_tokens']) if args.tokenize_in_workers else None
        train_loader = create_loader([train_dataset], samplers[:1], batch_size=[config['batch_size_train']],
                                     num_workers=[4], is_trains=[True], collate_fns=[train_collate])[0]
        test_loaders = {k: create_loader([v], [None], batch_size=[config['batch_size_test']], num_workers=[4],
                                         is_trains=[False], collate_fns=[None])[0] for k, v in test_dataset_dict.items()}
        tokenizer = mbert_build_tokenizer()

    print('Creating model', flush=True)
    torch.manual_seed(args.seed)                       # identical initial weights on every rank (DDP's broadcast)
    model = RetrievalModel(config=config)
    model.init_caption_encoder()
    model = model.to(device)
    if args.checkpoint != 'null':
        checkpoint = torch.load(args.checkpoint, map_location='cpu')
        print(model.load_state_dict(checkpoint['model'], strict=False))

    arg_opt = config['optimizer']
    optimizer = attach_grad_sync(create_optimizer(arg_opt, model))      # replaces the DDP wrapper (arena all-reduce)
    optimizer.enable_backward_fusion()          # train() calls step() after every backward(): non-ViT parameters update during the ViT backward
    arg_sche = dict(config['schedular'])
    arg_sche['step_per_epoch'] = max(1, len(train_loader))
    lr_scheduler = create_scheduler(arg_sche, optimizer)

    best, best_epoch = 0, 0
    start_time = time.time()
    for epoch in range(0, config['schedular']['epochs']):
        train_stats = {} if args.evaluate else train(model, train_loader, optimizer, tokenizer, epoch, device,
                                                     lr_scheduler, config)
        log_stats = {**{f'train_{k}': v for k, v in train_stats.items()}, 'epoch': epoch}
        for language, test_loader in test_loaders.items():
            score_i2t, score_t2i = evaluation_coarse(model, test_loader, tokenizer, device, config)
            if utils.is_main_process():
                res = itm_eval(score_i2t, score_t2i, test_loader.dataset.txt2img, test_loader.dataset.img2txt)
                print(language, res)
                log_stats.update({f'test_{language}_{k}': float(v) for k, v in res.items()})
                if res['sumr_sum'] > best:
                    best, best_epoch = res['sumr_sum'], epoch
                    torch.save({'model': model.state_dict(), 'optimizer': optimizer.state_dict(),
                                'lr_scheduler': lr_scheduler.state_dict(), 'config': config, 'epoch': epoch},
                               os.path.join(args.output_dir, 'checkpoint_best.pth'))
            if utils.is_dist_avail_and_initialized():
                dist.barrier()
        if utils.is_main_process():
            with open(os.path.join(args.output_dir, 'log.txt'), 'a') as f:
                f.write(json.dumps(log_stats) + '\n')
            if epoch == config['schedular']['epochs'] - 1 and not args.evaluate:
                torch.save({'model': model.state_dict(), 'optimizer': optimizer.state_dict(),
                            'lr_scheduler': lr_scheduler.state_dict(), 'config': config, 'epoch': epoch},
                           os.path.join(args.output_dir, f'checkpoint_{epoch}.pth'))
        if args.evaluate:
            break
    print('Training time {}'.format(str(datetime.timedelta(seconds=int(time.time() - start_time)))))
    if utils.is_main_process():
        print(f'best epoch: {best_epoch}')
    return model


def build_parser():
    """Flags of image_Retrieval_caption.py:519-534 (+ --synthetic / --synthetic_steps)."""
    p = argparse.ArgumentParser()
    p.add_argument('--checkpoint', type=str, default='null')
    p.add_argument('--config', type=str, default=None)
    p.add_argument('--output_dir', type=str, default='output/itr')
    p.add_argument('--device', default='cuda')
    p.add_argument('--seed', default=42, type=int)
    p.add_argument('--world_size', default=1, type=int, help='number of distributed processes')
    p.add_argument('--dist_url', default='env://', help='url used to set up distributed training')
    p.add_argument('--distributed', action='store_false')
    p.add_argument('--epoch', default=-1, type=int)
    p.add_argument('--bs', default=-1, type=int, help='for each gpu, batch_size = bs // num_gpus')
    p.add_argument('--evaluate', action='store_true')
    p.add_argument('--synthetic', action='store_true')
    p.add_argument('--tokenize_in_workers', action='store_true', help='tokenise in the DataLoader workers instead of in the train loop')
    p.add_argument('--synthetic_steps', default=4, type=int)
    return p


def load_config(args):
    if args.config:
        import yaml
        return yaml.load(open(args.config, 'r'), Loader=yaml.Loader)
    cfg = dict(MULTI30K_CFG)
    cfg.update(batch_size_train=32, batch_size_test=6, batch_size_test_text=32, max_tokens=200)
    return cfg


if __name__ == '__main__':
    args = build_parser().parse_args()
    config = load_config(args)
    Path(args.output_dir).mkdir(parents=True, exist_ok=True)
    import yaml
    yaml.dump(config, open(os.path.join(args.output_dir, 'config.yaml'), 'w'))
    main(args, config)
