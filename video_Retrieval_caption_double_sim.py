#!/usr/bin/env python3
"""Counterpart of the reference's LECCR/video_Retrieval_caption_double_sim.py (same CLI and train / evaluation_coarse /
itm_eval / main structure) running on leccr_amd: frame features + frame mask in, the video RetrievalModel
(leccr_amd/models/video_model_retrieval_caption.py), "double sim" evaluation.

    python3 -m torch.distributed.run --nproc_per_node=4 video_Retrieval_caption_double_sim.py --config cfg.yaml --bs 256
    python3 video_Retrieval_caption_double_sim.py --synthetic --bs 32 --epoch 1      # seeded VATEX-shaped batches

Real data needs the reference's `dataset` package (BigFile frame features are read by leccr_amd.bigfile.BigFile).
"""
import datetime
import json
import os
import time
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

import image_Retrieval_caption as image_task
from leccr_amd import utils
from leccr_amd.evaluation import evaluation_coarse_video as evaluation_coarse, itm_eval      # noqa: F401
from leccr_amd.models.video_model_retrieval_caption import RetrievalModel
from leccr_amd.optim import create_optimizer
from leccr_amd.parallel import attach_grad_sync
from leccr_amd.pipeline import DeferredScalars, DevicePrefetcher, TokenizingCollate, tokenize_or_pass
from leccr_amd.scheduler import create_scheduler
from leccr_amd.synthetic import VATEX_CFG, make_video_batch


def train(model, data_loader, optimizer, tokenizer, epoch, device, scheduler, config):
    """video_Retrieval_caption_double_sim.py:30-92, with the batch upload and loss logging of leccr_amd.pipeline."""
    model.train()
    metric_logger = utils.MetricLogger(delimiter='  ')
    metric_logger.add_meter('lr', utils.SmoothedValue(window_size=1, fmt='{value:.6f}'))
    names = ('loss_itc_vs', 'loss_itc_vt', 'loss_itc_st', 'loss_itc_c', 'loss_reg_c')
    for name in names:
        metric_logger.add_meter(name, utils.SmoothedValue(window_size=1, fmt='{value:.4f}'))
    header = 'Train Epoch: [{}]'.format(epoch)
    deferred = DeferredScalars()
    batches = DevicePrefetcher(data_loader, device)
    for i, (video, video_mask, text_ls, captions, idx, cap_idx) in enumerate(metric_logger.log_every(batches, 50, header)):
        # tokenised in the loader workers (--tokenize_in_workers: TokenizingCollate) or here, as the reference does
        text_inputs = [tokenize_or_pass(tokenizer, text, config['max_tokens'], device) for text in text_ls]
        captions = tokenize_or_pass(tokenizer, captions, config['max_tokens'], device)
        losses = model(video, video_mask, text_inputs, captions=captions, idx=idx, epoch=epoch, cap_idx=cap_idx)
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
            d = make_video_batch(self.b, rank=self.rank, seed=42 + 1000 * s, **self.g)
            yield d['video'], d['video_mask'], d['text'], d['captions'], d['idx'], None


def main(args, config):
    utils.init_distributed_mode(args)
    device = torch.device(args.device)
    world_size = utils.get_world_size()
    if args.epoch > 0:
        config['schedular']['epochs'] = args.epoch
    if args.bs > 0:
        config['batch_size_train'] = args.bs // world_size
    seed = args.seed + utils.get_rank()
    torch.manual_seed(seed)
    np.random.seed(seed)
    config['seed'] = args.seed
    if args.synthetic:
        geo = dict(frames=config.get('synthetic_frames', 8), width=config['vision_width'],
                   L_t=config.get('synthetic_L_t', 32), L_c=config.get('synthetic_L_c', 64),
                   vocab=config.get('synthetic_vocab', 119547))
        train_loader = _SyntheticTrain(args.synthetic_steps, config['batch_size_train'], utils.get_rank(), geo)
        tokenizer = image_task._PassThroughTokenizer()
        test_loaders = {}
    else:
        from dataset import create_dataset, create_sampler, create_loader, mbert_build_tokenizer   # reference package
        train_dataset, val_dataset_dict, test_dataset_dict = create_dataset('re_video_caption', config, args.evaluate)
        samplers = create_sampler([train_dataset], [True], world_size, utils.get_rank()) + [None]
        # the reference pads the ragged frame lists in the loader's collate_fn (video_Retrieval_caption_double_sim.py:288-292);
        # --tokenize_in_workers additionally runs the step's three tokenizer calls there (pipeline.TokenizingCollate)
        from dataset.retrieval_dataset_video import collate_fn
        tokenizer = mbert_build_tokenizer()
        train_collate = TokenizingCollate(tokenizer, config['max_tokens'], base_collate=collate_fn, text_pos=2) \
            if args.tokenize_in_workers else collate_fn
        train_loader = create_loader([train_dataset], samplers[:1], batch_size=[config['batch_size_train']],
                                     num_workers=[4], is_trains=[True], collate_fns=[train_collate])[0]
        test_loaders = {k: create_loader([v], [None], batch_size=[config['batch_size_test']], num_workers=[4],
                                         is_trains=[False], collate_fns=[None])[0] for k, v in test_dataset_dict.items()}
        tokenizer = mbert_build_tokenizer()

    print('Creating model', flush=True)
    torch.manual_seed(args.seed)
    model = RetrievalModel(config=config)
    model.init_caption_encoder()
    model = model.to(device)
    if args.checkpoint != 'null':
        checkpoint = torch.load(args.checkpoint, map_location='cpu')
        print(model.load_state_dict(checkpoint['model'], strict=False))
    optimizer = attach_grad_sync(create_optimizer(config['optimizer'], model))
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
        if args.evaluate:
            break
    print('Training time {}'.format(str(datetime.timedelta(seconds=int(time.time() - start_time)))))
    if utils.is_main_process():
        print(f'best epoch: {best_epoch}')
    return model


def load_config(args):
    if args.config:
        import yaml
        return yaml.load(open(args.config, 'r'), Loader=yaml.Loader)
    cfg = dict(VATEX_CFG)
    cfg.update(batch_size_train=32, batch_size_test=6, batch_size_test_text=32, max_tokens=200)
    return cfg


if __name__ == '__main__':
    args = image_task.build_parser().parse_args()
    config = load_config(args)
    Path(args.output_dir).mkdir(parents=True, exist_ok=True)
    import yaml
    yaml.dump(config, open(os.path.join(args.output_dir, 'config.yaml'), 'w'))
    main(args, config)
