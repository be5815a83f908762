"""create_optimizer with the reference's signature and grouping rules (optim.py:8-65), executed as ONE fused AdamW
launch over the ParamArena (leccr_adamw) instead of ~600 per-tensor update chains.

Semantics kept from the reference: transformers.AdamW update rule (bias correction on, decay applied to the updated
value, eps outside the bias correction), betas (0.9, 0.98), eps 1e-8, the substring-based no-decay set (so CLIP's
ln_*.weight DO get weight decay - 'ln_1.weight' matches none of the substrings), parameters that never receive a
gradient are skipped entirely.  model.init_params is [] for RetrievalModel (model_retrieval_caption.py:14), so the two
lr*lr_mult groups are empty; a non-empty init_params is rejected rather than silently ignored.
"""
import torch

NO_DECAY = ('bias', 'LayerNorm.bias', 'LayerNorm.weight', 'norm.bias', 'norm.weight', 'norm1.bias', 'norm1.weight',
            'norm2.bias', 'norm2.weight')


def decay_of_factory(wd):
    def decay_of(name, p):
        return 0.0 if any(nd in name for nd in NO_DECAY) else wd
    return decay_of


class ArenaAdamW(torch.optim.Optimizer):
    """torch.optim.Optimizer facade (param_groups / LambdaLR compatible) over ParamArena.adamw_step."""

    def __init__(self, model, lr, weight_decay, betas=(0.9, 0.98), eps=1e-8):
        self.model = model
        arena = model._ensure_runtime()
        super().__init__([arena.data], dict(lr=lr, weight_decay=weight_decay, betas=betas, eps=eps))
        self.arena = arena
        self._decay_of = decay_of_factory(weight_decay)
        self.grad_sync = None          # set by leccr_amd.parallel for W > 1
        self.grad_scale = 1.0          # 1/W under data parallelism (mean over ranks, folded into the AdamW launch)

    def zero_grad(self, set_to_none=False):
        self.arena.zero_grad()

    @torch.no_grad()
    def step(self, closure=None):
        g = self.param_groups[0]
        from . import ops
        ops.join_wgrad_stream()
        if self.grad_sync is not None:
            self.grad_sync()
        self.arena.mark_autograd_grads()
        self.arena.adamw_step(g['lr'], g['betas'], g['eps'], self._decay_of, self.grad_scale)
        self.arena.refresh(adamw_wrote_lp=True)

    def state_dict(self):
        a = self.arena
        return {'step': a.step_count, 'm': a.state_m, 'v': a.state_v,
                'param_groups': [{k: v for k, v in g.items() if k != 'params'} for g in self.param_groups]}

    def load_state_dict(self, sd):
        a = self.arena
        a.step_count = sd['step']
        if sd['m'] is not None:
            a.state_m, a.state_v = sd['m'].to(a.data.device).clone(), sd['v'].to(a.data.device).clone()
        for g, s in zip(self.param_groups, sd['param_groups']):
            g.update(s)


def create_optimizer(args, model):
    """optim.py:8-65.  args: .lr, .weight_decay, optional .lr_mult."""
    get = (lambda k, d=None: args.get(k, d)) if isinstance(args, dict) else (lambda k, d=None: getattr(args, k, d))
    lr, wd = get('lr'), get('weight_decay')
    print('### lr_mult, ', get('lr_mult', 1))
    large_lr = getattr(model, 'init_params', [])
    print("### model has 'init_params', ", len(large_lr))
    if len(large_lr):
        raise NotImplementedError('per-parameter lr multipliers are unused by the reference model (init_params == [])')
    return ArenaAdamW(model, lr=lr, weight_decay=wd, betas=(0.9, 0.98), eps=1e-8)
