"""create_scheduler with the reference's signature (scheduler.py:4-28): linear warm-up then linear decay LambdaLR."""
from torch.optim.lr_scheduler import LambdaLR


class _Args(dict):
    __getattr__ = dict.get

    def __setattr__(self, k, v):
        self[k] = v


def create_scheduler(args, optimizer):
    args = args if hasattr(args, 'sched') and not isinstance(args, dict) else _Args(args)
    if 'num_training_steps' not in args:
        args['num_training_steps'] = args['epochs'] * args['step_per_epoch']
    print('### num_training_steps, ', args['num_training_steps'], flush=True)
    if isinstance(args['num_warmup_steps'], float):
        assert 0 <= args['num_warmup_steps'] < 1
        args['num_warmup_steps'] = int(args['num_training_steps'] * args['num_warmup_steps'])
    print('### num_warmup_steps, ', args['num_warmup_steps'], flush=True)
    if args['sched'] != 'linear':
        raise NotImplementedError(f"args.sched == {args['sched']}")
    warm, total = args['num_warmup_steps'], args['num_training_steps']

    def lr_lambda(current_step):
        if current_step < warm:
            return float(current_step) / float(max(1, warm))
        return max(0.0, float(total - current_step) / float(max(1, total - warm)))

    return LambdaLR(optimizer, lr_lambda, last_epoch=-1)
