"""Parameter-holder modules with the reference's attribute names (so state_dict keys match key for key)."""
import math

import torch
import torch.nn as nn


class Linear(nn.Module):
    """Holds weight [out,in] (+ bias) like nn.Linear; the compute is a HIP GEMM (leccr_amd.ops.linear)."""

    def __init__(self, in_features, out_features, bias=True, init='linear'):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        self.bias = nn.Parameter(torch.zeros(out_features)) if bias else None
        if init == 'linear':        # nn.Linear default
            nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
            if bias:
                bound = 1 / math.sqrt(in_features)
                nn.init.uniform_(self.bias, -bound, bound)
        elif init == 'bert':        # HF BERT: N(0, 0.02), zero bias
            nn.init.normal_(self.weight, std=0.02)

    def forward(self, x, wgrad_scale=1.0, out_f32=False):
        from .. import ops
        return ops.linear(x, self.weight, self.bias, wgrad_scale, out_f32)


class LayerNormParams(nn.Module):
    def __init__(self, d, eps):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(d))
        self.bias = nn.Parameter(torch.zeros(d))
        self.eps = eps

    def forward(self, x):
        from .. import ops
        return ops.layer_norm(x, self.weight, self.bias, self.eps)


def mark_unused(module):
    """Parameters the reference allocates but never trains (no gradient ever reaches them): kept for state_dict
    compatibility, excluded from the ParamArena (no compute copies, no optimizer state)."""
    for p in module.parameters():
        p._leccr_unused = True
    return module
