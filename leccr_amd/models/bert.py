"""BERT encoder with HF BertModel's module tree / state_dict keys (the reference's text tower: models/xvlm.py:91-103,
called at xvlm.py:238 and model_retrieval_caption.py:60) over the HIP ops.

Arithmetic restated from transformers' modeling_bert (third-party; `transformers==4.12.5` in LECCR/requirements.txt):
embeddings = word[ids] + position[0..L-1] + token_type[0] -> LayerNorm(1e-12); post-LN layers; erf-GELU; additive
key-padding mask; dropout (hidden 0.1, attention probabilities 0.1 - xvlm.py:94-97) is active in train mode with
counter-based masks regenerated in the backward (leccr_dropout, leccr_attn_* drop_p).
"""
import types

import torch
import torch.nn as nn

from .. import ops
from .layers import LayerNormParams, Linear, mark_unused

BERT_BASE_MULTILINGUAL = dict(vocab_size=119547, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                              intermediate_size=3072, max_position_embeddings=512, type_vocab_size=2,
                              layer_norm_eps=1e-12, pad_token_id=0,
                              hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)     # xvlm.py:94-97


class BertEmbeddings(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.word_embeddings = nn.Embedding(c['vocab_size'], c['hidden_size'], padding_idx=c['pad_token_id'])
        self.position_embeddings = nn.Embedding(c['max_position_embeddings'], c['hidden_size'])
        self.token_type_embeddings = nn.Embedding(c['type_vocab_size'], c['hidden_size'])
        self.LayerNorm = LayerNormParams(c['hidden_size'], c['layer_norm_eps'])
        self.eps, self.pad_id = c['layer_norm_eps'], c['pad_token_id']
        self.dropout_p = c['hidden_dropout_prob']
        for e in (self.word_embeddings, self.position_embeddings, self.token_type_embeddings):
            nn.init.normal_(e.weight, std=0.02)
        with torch.no_grad():
            self.word_embeddings.weight[self.pad_id].zero_()

    def _params(self):
        return (self.word_embeddings.weight, self.position_embeddings.weight, self.token_type_embeddings.weight,
                self.LayerNorm.weight, self.LayerNorm.bias)


class _SelfAttn(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.query, self.key, self.value = Linear(d, d, init='bert'), Linear(d, d, init='bert'), Linear(d, d, init='bert')


class _SelfOutput(nn.Module):
    def __init__(self, d_in, d, eps):
        super().__init__()
        self.dense = Linear(d_in, d, init='bert')
        self.LayerNorm = LayerNormParams(d, eps)


class _Attention(nn.Module):
    def __init__(self, d, eps):
        super().__init__()
        self.self = _SelfAttn(d)
        self.output = _SelfOutput(d, d, eps)


class _Intermediate(nn.Module):
    def __init__(self, d, ffn):
        super().__init__()
        self.dense = Linear(d, ffn, init='bert')


class BertLayer(nn.Module):
    def __init__(self, c):
        super().__init__()
        d = c['hidden_size']
        self.heads, self.eps = c['num_attention_heads'], c['layer_norm_eps']
        self.fp8 = False               # set by XVLMBase._ensure_runtime for compute_dtype 'fp8'
        self.hidden_dropout_p, self.attn_dropout_p = c['hidden_dropout_prob'], c['attention_probs_dropout_prob']
        self.attention = _Attention(d, self.eps)
        self.intermediate = _Intermediate(d, c['intermediate_size'])
        self.output = _SelfOutput(c['intermediate_size'], d, self.eps)

    def _arena_groups(self):
        """query/key/value live adjacently in the arena -> one packed [3d,d] QKV operand (one GEMM, one dW GEMM)."""
        s = self.attention.self
        return [('_qkv_w', [s.query.weight, s.key.weight, s.value.weight]),
                ('_qkv_b', [s.query.bias, s.key.bias, s.value.bias])]

    def _params(self):
        s, ao = self.attention.self, self.attention.output
        return (s.query.weight, s.query.bias, s.key.weight, s.key.bias, s.value.weight, s.value.bias,
                ao.dense.weight, ao.dense.bias, ao.LayerNorm.weight, ao.LayerNorm.bias,
                self.intermediate.dense.weight, self.intermediate.dense.bias,
                self.output.dense.weight, self.output.dense.bias, self.output.LayerNorm.weight, self.output.LayerNorm.bias)

    def forward(self, x, kmask):
        return ops.BertLayerFn.apply(x, kmask, self, *self._params())


class _Encoder(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.layer = nn.ModuleList([BertLayer(c) for _ in range(c['num_hidden_layers'])])


class _Pooler(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.dense = Linear(d, d, init='bert')


class BertModel(nn.Module):
    """`BertModel(ids, attention_mask=mask, return_dict=True).last_hidden_state` - the only call the path makes."""

    def __init__(self, config=None, **overrides):
        super().__init__()
        c = dict(BERT_BASE_MULTILINGUAL)
        c.update(config or {})
        c.update(overrides)
        self.config = types.SimpleNamespace(**c)
        self.embeddings = BertEmbeddings(c)
        self.encoder = _Encoder(c)
        self.pooler = mark_unused(_Pooler(c['hidden_size']))        # allocated by HF, never used by LECCR
        self.compute_dtype = torch.float32

    def forward(self, input_ids, attention_mask=None, return_dict=True, **_):
        b, L = input_ids.shape
        if attention_mask is None:
            attention_mask = torch.ones_like(input_ids)
        kmask = (attention_mask == 0).to(torch.uint8).contiguous()
        x = ops.BertEmbedFn.apply(input_ids, self.embeddings, self.compute_dtype, *self.embeddings._params())
        for layer in self.encoder.layer:
            x = layer(x, kmask)
        return types.SimpleNamespace(last_hidden_state=x)
