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
        self.backward_fusion = False   # enable_backward_fusion(): update the non-ViT parameters during the ViT backward
        self._early_handles = []       # W > 1: the early region's all-reduce handles (leccr_amd.parallel)
        self._early_updated = False
        self._update_stream = None

    def enable_backward_fusion(self):
        """Update the arena's early region (text tower, heads, interaction: every parameter whose gradient is final when the ViT
        backward starts) on a stream of its own WHILE the ViT backward runs, instead of after it: an HBM-bound pass beside
        MFMA-bound GEMMs.  Only for loops in which every backward() is followed by step() (the reference's train loop;
        bench.py) - a backward() without step() would leave those parameters updated."""
        self.backward_fusion = True
        self.arena.on_early_grads_final = self._early_update      # keyed on this model's arena (ops._fire)
        return self

    @torch.no_grad()
    def _early_update(self):
        a = self.arena
        split = a.early_numel
        if not self.backward_fusion or self._early_updated or not (0 < split < a.numel) or not a.data.is_cuda:
            return
        from . import diag, ops
        g = self.param_groups[0]
        main = torch.cuda.current_stream()
        with diag.host('early_update'):
            self._early_update_launch(a, split, g, main, ops)
        self._early_updated = True

    def _early_update_launch(self, a, split, g, main, ops):
        us = self._update_stream = ops.update_stream(a.data.device)
        a.mark_autograd_grads(0, split)
        us.wait_stream(main)                    # interaction / heads / losses backward
        ops.join_wgrad_stream(us)               # text-tower backward and every weight-gradient launch so far
        with torch.cuda.stream(us):
            for w in self._early_handles:       # W > 1: the region's all-reduce
                w.wait()
            a.adamw_step(g['lr'], g['betas'], g['eps'], self._decay_of, self.grad_scale, 0, split, step=a.step_count + 1)
            a.refresh(adamw_wrote_lp=True, lo=0, hi=split)

    def zero_grad(self, set_to_none=False):
        self.arena.zero_grad()

    @torch.no_grad()
    def step(self, closure=None):
        g = self.param_groups[0]
        from . import ops
        ops.join_wgrad_stream()
        if self.grad_sync is not None:
            self.grad_sync()
        a = self.arena
        if self._early_updated:                 # the early region was updated during the ViT backward (same step number)
            self._early_updated = False
            self._early_handles = []
            torch.cuda.current_stream().wait_stream(self._update_stream)
            split = a.early_numel
            a.mark_autograd_grads(split, a.numel)
            a.adamw_step(g['lr'], g['betas'], g['eps'], self._decay_of, self.grad_scale, split, a.numel)
            a.refresh(adamw_wrote_lp=True, lo=split, hi=a.numel)
            return
        self._early_handles = []
        a.mark_autograd_grads()
        a.adamw_step(g['lr'], g['betas'], g['eps'], self._decay_of, self.grad_scale)
        a.refresh(adamw_wrote_lp=True)

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
