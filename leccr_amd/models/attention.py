"""CrossAttention / SelfAttention stacks with the reference's module tree (models/attention.py:24-128).

Tensors are batch-major [b, L, d] here (the reference's LND is a layout choice of nn.MultiheadAttention); the
RetrievalModel transposes at its public boundary where the reference API returns LND.
"""
import copy

import torch
import torch.nn as nn

from .. import ops
from .clip_vit import _MHAParams
from .layers import LayerNormParams, Linear, mark_unused


class CrossAttentionLayer(nn.Module):
    def __init__(self, d_model, nhead, dropout=0.1, activation='gelu'):
        super().__init__()
        assert activation == 'gelu'
        self.heads = nhead
        self.multihead_attn = _MHAParams(d_model)
        self.linear1 = Linear(d_model, d_model)
        self.linear2 = Linear(d_model, d_model)
        self.norm1 = mark_unused(LayerNormParams(d_model, 1e-5))     # allocated, unused (attention.py:33,47-54)
        self.norm2 = LayerNormParams(d_model, 1e-5)
        self.dropout_p = dropout

    def _params(self):
        m = self.multihead_attn
        return (m.in_proj_weight, m.in_proj_bias, m.out_proj.weight, m.out_proj.bias, self.linear1.weight,
                self.linear1.bias, self.linear2.weight, self.linear2.bias, self.norm2.weight, self.norm2.bias)

    def forward(self, tgt, memory, memory_key_padding_mask=None):
        km = None
        if memory_key_padding_mask is not None:
            km = memory_key_padding_mask.to(torch.uint8).contiguous()
        return ops.CrossAttnLayerFn.apply(tgt, memory, km, self, *self._params())


class CrossAttention(nn.Module):
    def __init__(self, d_model, nhead, num_layers=1, dropout=0.1):
        super().__init__()
        layer = CrossAttentionLayer(d_model, nhead, dropout)
        self.layers = nn.ModuleList([copy.deepcopy(layer) for _ in range(num_layers)])   # clones share init values

    def forward(self, tgt, memory, memory_key_padding_mask=None):
        out = tgt
        for layer in self.layers:
            out = layer(out, memory, memory_key_padding_mask)
        return out


class SelfAttentionLayer(CrossAttentionLayer):
    """q = k = v = src (attention.py:87-113)."""

    def forward(self, src, memory_key_padding_mask=None):
        return CrossAttentionLayer.forward(self, src, src, memory_key_padding_mask)


class SelfAttention(nn.Module):
    def __init__(self, d_model, nhead, num_layers=1, dropout=0.1):
        super().__init__()
        layer = SelfAttentionLayer(d_model, nhead, dropout)
        self.layers = nn.ModuleList([copy.deepcopy(layer) for _ in range(num_layers)])

    def forward(self, src, memory_key_padding_mask=None):
        out = src
        for layer in self.layers:
            out = layer(out, memory_key_padding_mask)
        return out
