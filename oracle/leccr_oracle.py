"""CPU oracle for the LECCR retrieval hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A plain PyTorch fp32 (CPU) restatement of what one step of the reference computes, written
as functions over a flat {state_dict key: tensor} mapping.  It is the checker for the HIP path
in leccr_amd/: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.
leccr_amd never imports this package and fails loudly when its HIP library is missing.

Parity pinning: tests/test_oracle_golden.py checks every function here against fixtures captured
from the reference's own classes (tests/golden/make_golden.py, run in the build container):
ViT trunk, BERT, caption interaction, every loss (value + gradients), the full 5-loss forward with
gradients at W=1 and under 2-rank DDP, and itm_eval's integer ranks.  The text tower's arithmetic
is third-party (HF transformers BertModel, `transformers==4.12.5` per LECCR/requirements.txt:2;
the fixtures were captured with transformers 5.15.0) -- the reference ships no tests or golden
vectors of its own, so the captured fixtures are the only pin.

Each function cites the reference lines it restates (paths relative to /root/reference/LECCR).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------------------- bf16 storage model (optional)
# `with lowp():` makes the oracle round to bf16 wherever the MI355X throughput path materialises a tensor in bf16 (activations
# after every GEMM / LayerNorm / attention / residual add, the attention probabilities fed to the PV product, the bf16 compute
# copies of the weights, and - in backward - the gradient of each such tensor), keeping f32 accumulation, statistics, biases,
# gains, heads and losses, exactly like DESIGN.md section 2 describes the bf16 path.  It is NOT the reference's arithmetic (the
# reference is fp32 throughout, models/xvlm.py:88) and is off by default - every fixture check runs with it off.  The GPU
# tests use it as the NOISE MODEL of the bf16 path: how far bf16 storage at those points moves each loss / gradient away
# from the fp32 result is the yardstick the HIP bf16 path's own deviation is held against, tensor by tensor.
_LOWP = False
_FP8 = False
_FP8_DW = False


class lowp:
    """with lowp(): bf16 storage model.  with lowp(fp8=True): additionally the fp8 operand model of compute_dtype 'fp8' (BASELINE
    configs[4]): the big Linears of the ViT blocks and BERT layers multiply per-tensor scaled OCP e4m3 copies of their input and
    weight (power-of-two scale from the tensor's amax), and their dX product takes the incoming gradient in e5m2; with fp8_dw (round 3)
    the weight-gradient product dY^T X of those Linears takes the same e5m2 / e4m3 copies where the HIP path has an fp8 kernel for it
    (>= 2048 rows, in / out whole 256-tiles); everything else as under lowp()."""

    def __init__(self, fp8=False, fp8_dw=True):
        self.fp8, self.fp8_dw = fp8, fp8 and fp8_dw

    def __enter__(self):
        global _LOWP, _FP8, _FP8_DW
        self.prev, _LOWP, _FP8, _FP8_DW = (_LOWP, _FP8, _FP8_DW), True, self.fp8, self.fp8_dw

    def __exit__(self, *exc):
        global _LOWP, _FP8, _FP8_DW
        _LOWP, _FP8, _FP8_DW = self.prev


def _q8(t, dtype, fmax):
    """fp8 copy of t as the product sees it: round(t * s) / s with s = 2^floor(log2(fmax / amax))."""
    amax = float(t.detach().abs().max())
    if amax == 0.0:
        return t
    sc = 2.0 ** math.floor(math.log2(fmax / amax))
    return (t * sc).to(dtype).float() / sc


class _Fp8Linear(torch.autograd.Function):
    """y = q4(x) q4(w)^T + b;  dx = q5(dy) q4(w);  db = colsum(dy) from the bf16 tensor;  dw = dy^T x from the bf16 tensors, or
    q5(dy)^T q4(x) where the HIP path's fp8 weight-gradient kernel applies (lowp(fp8_dw=True); its < 128-row bf16 tail is not modelled)."""

    @staticmethod
    def forward(ctx, x, w, b):
        wr = w.bfloat16().float()
        wq = _q8(wr, torch.float8_e4m3fn, 448.0)
        ctx.save_for_backward(x, wq)
        ctx.has_b = b is not None
        y = F.linear(_q8(x, torch.float8_e4m3fn, 448.0), wq)
        return y + b if b is not None else y

    @staticmethod
    def backward(ctx, dy):
        x, wq = ctx.saved_tensors
        dy2, x2 = dy.reshape(-1, dy.shape[-1]), x.reshape(-1, x.shape[-1])
        dq = _q8(dy2, torch.float8_e5m2, 57344.0)
        dx = (dq @ wq).reshape(x.shape)
        if _FP8_DW and dy2.shape[0] >= 2048 and dy2.shape[1] % 256 == 0 and x2.shape[1] % 256 == 0:
            dw = dq.t() @ _q8(x2, torch.float8_e4m3fn, 448.0)
        else:
            dw = dy2.t() @ x2
        return dx, dw, (dy2.sum(0) if ctx.has_b else None)


class _RoundAct(torch.autograd.Function):
    """A tensor stored in bf16: rounded going forward, its gradient (also stored in bf16) rounded coming back."""

    @staticmethod
    def forward(ctx, x):
        return x.bfloat16().float()

    @staticmethod
    def backward(ctx, g):
        return g.bfloat16().float()


class _RoundWeight(torch.autograd.Function):
    """The bf16 compute copy of an f32 master weight: rounded going forward; its gradient accumulates in f32."""

    @staticmethod
    def forward(ctx, w):
        return w.bfloat16().float()

    @staticmethod
    def backward(ctx, g):
        return g


def _r(x):
    return _RoundAct.apply(x) if _LOWP else x


def _w(w):
    return _RoundWeight.apply(w) if _LOWP else w


def linear(x, w, b=None, keep_f32=False, fp8=False):
    """F.linear; under lowp() on the weight's bf16 copy, result stored in bf16 unless keep_f32 (the heads feeding the losses).
    fp8: an encoder-block Linear - under lowp(fp8=True) its product takes fp8 operands when the shape has an fp8 kernel
    (rows >= 256, out >= 256 and % 32 == 0, in % 128 == 0: leccr_gemm's LECCR_FP8 rule)."""
    rows = x.numel() // x.shape[-1]
    if _LOWP and _FP8 and fp8 and rows >= 256 and w.shape[0] >= 256 and w.shape[0] % 32 == 0 and w.shape[1] % 128 == 0:
        y = _Fp8Linear.apply(x, w, b)
    else:
        y = F.linear(x, _w(w), b)
    return y if keep_f32 else _r(y)


# --------------------------------------------------------------------------------------- primitives
def layer_norm(x, w, b, eps):
    """nn.LayerNorm over the last axis, statistics in fp32 (clip/model.py:157-163)."""
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return _r((x - mu) / torch.sqrt(var + eps) * w + b)


def quick_gelu(x):
    """clip/model.py:166-168."""
    return x * torch.sigmoid(1.702 * x)


def gelu_erf(x):
    """Exact GELU (HF BERT hidden_act='gelu'; attention.py:16-17 F.gelu)."""
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


class _FlashBF16(torch.autograd.Function):
    """lowp() only: the storage model of a bf16 flash attention (attention kernels).  Forward: f32
    scores from bf16 q / k, probabilities rounded to bf16 for the PV product, the normaliser summed from those rounded
    values, the output stored in bf16.  Backward: probabilities recomputed in f32 from the saved log-sum-exp, the softmax row
    constant taken as rowsum(dO * O) from the STORED (bf16) output - not as sum_j P_ij dP_ij - and dS / P rounded to bf16 as
    operands of the dQ / dK / dV products.  q, k, v: [b, h, L, hd] (bf16-representable values)."""

    @staticmethod
    def forward(ctx, q, k, v, mask, scale):
        s = (q @ k.transpose(-1, -2)) * scale
        if mask is not None:
            s = s.masked_fill(mask[:, None, None, :], float('-inf'))
        m = s.amax(-1, keepdim=True)
        e = torch.exp(s - m).bfloat16().float()
        l = e.sum(-1, keepdim=True)
        o = ((e @ v) / l).bfloat16().float()
        ctx.save_for_backward(q, k, v, o, m + torch.log(l))
        ctx.mask, ctx.scale = mask, scale
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, o, lse = ctx.saved_tensors
        do = do.bfloat16().float()
        s = (q @ k.transpose(-1, -2)) * ctx.scale
        if ctx.mask is not None:
            s = s.masked_fill(ctx.mask[:, None, None, :], float('-inf'))
        p = torch.exp(s - lse)
        dp = do @ v.transpose(-1, -2)
        delta = (do * o).sum(-1, keepdim=True)
        ds = (p * (dp - delta)).bfloat16().float()
        r = lambda t: t.bfloat16().float()
        return r((ds @ k) * ctx.scale), r((ds.transpose(-1, -2) @ q) * ctx.scale), r(p.bfloat16().float().transpose(-1, -2) @ do), None, None


def attention_core(q, k, v, heads, key_padding_mask=None, causal=False):
    """Scaled dot-product attention, batch-major.  q [b,Lq,d], k/v [b,Lk,d]; mask [b,Lk] True = ignore.

    Semantics of torch nn.MultiheadAttention / HF BertSelfAttention: scores/sqrt(head_dim), masked keys get
    -inf (BERT: finfo.min, same softmax result), softmax over keys, no dropout (eval / p=0).
    """
    b, Lq, d = q.shape
    Lk = k.shape[1]
    hd = d // heads
    qh = q.reshape(b, Lq, heads, hd).permute(0, 2, 1, 3)
    kh = k.reshape(b, Lk, heads, hd).permute(0, 2, 1, 3)
    vh = v.reshape(b, Lk, heads, hd).permute(0, 2, 1, 3)
    s = (qh @ kh.transpose(-1, -2)) / math.sqrt(hd)
    if key_padding_mask is not None:
        s = s.masked_fill(key_padding_mask[:, None, None, :], float('-inf'))
    if causal:              # CLIP.build_attention_mask (clip/model.py:328-334): -inf above the diagonal
        s = s.masked_fill(torch.ones(Lq, Lk, dtype=torch.bool).triu(1), float('-inf'))
    if _LOWP and not causal:
        o = _FlashBF16.apply(qh, kh, vh, key_padding_mask, 1.0 / math.sqrt(hd))
    else:
        p = torch.softmax(s, dim=-1)
        o = p @ vh
    return o.permute(0, 2, 1, 3).reshape(b, Lq, d)


def torch_mha(sd, pre, query, memory, heads, key_padding_mask=None, keep_f32=False, fp8=False, causal=False):
    """nn.MultiheadAttention with packed in_proj ordered q,k,v (clip/model.py:175,187; attention.py:27,47).
    Batch-major in/out."""
    w, bias = sd[pre + 'in_proj_weight'], sd[pre + 'in_proj_bias']
    d = w.shape[1]
    if fp8 and query is memory:      # the ViT block's packed in_proj: one product, one weight scale (as the HIP path's packed GEMM)
        qkv = linear(query, w, bias, fp8=True)
        q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
    else:
        q = linear(query, w[:d], bias[:d])
        k = linear(memory, w[d:2 * d], bias[d:2 * d])
        v = linear(memory, w[2 * d:], bias[2 * d:])
    o = attention_core(q, k, v, heads, key_padding_mask, causal=causal)
    return linear(o, sd[pre + 'out_proj.weight'], sd[pre + 'out_proj.bias'], keep_f32=keep_f32, fp8=fp8)


# --------------------------------------------------------------------------------------- CLIP ViT
def vit_trunk(sd, pre, image):
    """VisionTransformer.forward (clip/model.py:223-240): patch conv -> CLS + pos -> ln_pre -> blocks.
    Returns [b, N, d] (batch-major; the reference's LND permutes are layout only)."""
    w = sd[pre + 'conv1.weight']
    d, _, p, _ = w.shape
    b = image.shape[0]
    g = image.shape[-1] // p
    # stride == kernel: the conv is a GEMM over non-overlapping patches
    image = image[..., :g * p, :g * p]          # conv1 has stride p and no padding: R % p edge pixels are never read
    patches = image.reshape(b, 3, g, p, g, p).permute(0, 2, 4, 1, 3, 5).reshape(b, g * g, 3 * p * p)
    x = _r(_r(patches) @ _w(w.reshape(d, -1)).t())
    cls = sd[pre + 'class_embedding'].reshape(1, 1, d).expand(b, 1, d)
    pos = sd[pre + 'positional_embedding']
    x = _r(torch.cat([cls + pos[:1], x + pos[1:]], 1))
    x = layer_norm(x, sd[pre + 'ln_pre.weight'], sd[pre + 'ln_pre.bias'], 1e-5)
    heads = d // 64
    i = 0
    while f'{pre}transformer.resblocks.{i}.ln_1.weight' in sd:
        bp = f'{pre}transformer.resblocks.{i}.'
        # ResidualAttentionBlock (clip/model.py:171-192): pre-LN attention + pre-LN QuickGELU MLP
        h = layer_norm(x, sd[bp + 'ln_1.weight'], sd[bp + 'ln_1.bias'], 1e-5)
        x = _r(x + torch_mha(sd, bp + 'attn.', h, h, heads, keep_f32=True, fp8=True))      # residual added in the GEMM epilogue (f32)
        h = layer_norm(x, sd[bp + 'ln_2.weight'], sd[bp + 'ln_2.bias'], 1e-5)
        h = _r(quick_gelu(linear(h, sd[bp + 'mlp.c_fc.weight'], sd[bp + 'mlp.c_fc.bias'], keep_f32=True, fp8=True)))
        x = _r(x + linear(h, sd[bp + 'mlp.c_proj.weight'], sd[bp + 'mlp.c_proj.bias'], keep_f32=True, fp8=True))
        i += 1
    return x


def encode_text_hidden(sd, pre, text):
    """CLIP.encode_text(return_hidden=True)[-1] (clip/model.py:348-365): token + positional embedding -> transformer with the causal
    mask -> ln_final on all tokens @ text_projection.  pre = 'clip_encoder.'; heads = width // 64 as CLIP.__init__ receives them from
    build_model (clip/model.py:427-429)."""
    x = _r(sd[pre + 'token_embedding.weight'][text] + sd[pre + 'positional_embedding'])
    d = x.shape[-1]
    heads = max(1, d // 64)
    i = 0
    while f'{pre}transformer.resblocks.{i}.ln_1.weight' in sd:
        bp = f'{pre}transformer.resblocks.{i}.'
        h = layer_norm(x, sd[bp + 'ln_1.weight'], sd[bp + 'ln_1.bias'], 1e-5)
        x = _r(x + torch_mha(sd, bp + 'attn.', h, h, heads, keep_f32=True, causal=True))
        h = layer_norm(x, sd[bp + 'ln_2.weight'], sd[bp + 'ln_2.bias'], 1e-5)
        h = _r(quick_gelu(linear(h, sd[bp + 'mlp.c_fc.weight'], sd[bp + 'mlp.c_fc.bias'], keep_f32=True)))
        x = _r(x + linear(h, sd[bp + 'mlp.c_proj.weight'], sd[bp + 'mlp.c_proj.bias'], keep_f32=True))
        i += 1
    return _r(layer_norm(x, sd[pre + 'ln_final.weight'], sd[pre + 'ln_final.bias'], 1e-5) @ _w(sd[pre + 'text_projection']))


def encode_image(sd, pre, image):
    """CLIP.encode_image(return_hidden=True) (clip/model.py:340-346): ln_post on ALL tokens, then @ proj."""
    x = vit_trunk(sd, pre, image)
    return _r(layer_norm(x, sd[pre + 'ln_post.weight'], sd[pre + 'ln_post.bias'], 1e-5) @ _w(sd[pre + 'proj']))


# --------------------------------------------------------------------------------------- BERT
def bert_forward(sd, pre, ids, mask, heads):
    """HF BertModel(ids, attention_mask).last_hidden_state as called at models/xvlm.py:238.

    word + position(0..L-1) + token_type(0) -> LayerNorm(1e-12); 12x post-LN layers with erf-GELU; additive
    key mask only (padded query rows are still computed).  The word table is nn.Embedding(padding_idx=
    pad_token_id=0): row 0 is read like any row but receives NO gradient.
    """
    L = ids.shape[1]
    x = F.embedding(ids, sd[pre + 'embeddings.word_embeddings.weight'], padding_idx=0) \
        + sd[pre + 'embeddings.position_embeddings.weight'][:L] \
        + sd[pre + 'embeddings.token_type_embeddings.weight'][0]
    x = layer_norm(_r(x), sd[pre + 'embeddings.LayerNorm.weight'], sd[pre + 'embeddings.LayerNorm.bias'], 1e-12)
    kpm = ~mask.bool()
    i = 0
    while f'{pre}encoder.layer.{i}.attention.self.query.weight' in sd:
        lp = f'{pre}encoder.layer.{i}.'
        if _LOWP and _FP8:      # the HIP path runs the three projections as one packed [3d, d] product: one weight scale
            wqkv = torch.cat([sd[lp + f'attention.self.{n}.weight'] for n in ('query', 'key', 'value')], 0)
            bqkv = torch.cat([sd[lp + f'attention.self.{n}.bias'] for n in ('query', 'key', 'value')], 0)
            q, k, v = linear(x, wqkv, bqkv, fp8=True).split(x.shape[-1], -1)
        else:
            q = linear(x, sd[lp + 'attention.self.query.weight'], sd[lp + 'attention.self.query.bias'])
            k = linear(x, sd[lp + 'attention.self.key.weight'], sd[lp + 'attention.self.key.bias'])
            v = linear(x, sd[lp + 'attention.self.value.weight'], sd[lp + 'attention.self.value.bias'])
        a = attention_core(q, k, v, heads, kpm)
        a = linear(a, sd[lp + 'attention.output.dense.weight'], sd[lp + 'attention.output.dense.bias'], keep_f32=True, fp8=True)
        x = layer_norm(_r(x + a), sd[lp + 'attention.output.LayerNorm.weight'],
                       sd[lp + 'attention.output.LayerNorm.bias'], 1e-12)
        h = _r(gelu_erf(linear(x, sd[lp + 'intermediate.dense.weight'], sd[lp + 'intermediate.dense.bias'], keep_f32=True, fp8=True)))
        h = linear(h, sd[lp + 'output.dense.weight'], sd[lp + 'output.dense.bias'], keep_f32=True, fp8=True)
        x = layer_norm(_r(x + h), sd[lp + 'output.LayerNorm.weight'], sd[lp + 'output.LayerNorm.bias'], 1e-12)
        i += 1
    return x


# --------------------------------------------------------------------------------------- caption interaction
def cross_attention(sd, pre, tgt, memory, key_padding_mask=None, heads=8):
    """CrossAttention / CrossAttentionLayer (models/attention.py:24-83), batch-major.

    Per layer: a = MHA(tgt, memory); out = norm2(tgt + linear2(gelu(linear1(a)))).  norm1 is unused and the
    attention output is NOT residual-added before the FFN (attention.py:47-54).
    """
    i = 0
    while f'{pre}layers.{i}.linear1.weight' in sd:
        lp = f'{pre}layers.{i}.'
        a = torch_mha(sd, lp + 'multihead_attn.', tgt, memory, heads, key_padding_mask)
        h = _r(gelu_erf(linear(a, sd[lp + 'linear1.weight'], sd[lp + 'linear1.bias'], keep_f32=True)))
        h = linear(h, sd[lp + 'linear2.weight'], sd[lp + 'linear2.bias'], keep_f32=True)
        tgt = layer_norm(_r(tgt + h), sd[lp + 'norm2.weight'], sd[lp + 'norm2.bias'], 1e-5)
        i += 1
    return tgt


def interaction_with_caption(sd, image_embeds, caption_hidden, key_padding_mask):
    """RetrievalModel.interaction_with_caption (models/model_retrieval_caption.py:76-84).
    All tensors batch-major here: returns (after_image [b,N,vw], after_caption [b,n,vw], ori_caption [b,n,vw])."""
    b = image_embeds.shape[0]
    queries = _r(sd['queries'].permute(1, 0, 2).expand(b, -1, -1))
    cap = linear(caption_hidden, sd['caption_proj.weight'], sd['caption_proj.bias'])
    ori_caption = cross_attention(sd, 'crossattn_query.', queries, cap, key_padding_mask)
    after_image = cross_attention(sd, 'crossattn.', image_embeds, ori_caption)
    after_caption = cross_attention(sd, 'crossattn2.', ori_caption, image_embeds)
    return after_image, after_caption, ori_caption


# --------------------------------------------------------------------------------------- all-gather
class _SliceBackwardGather(torch.autograd.Function):
    """AllGather (models/xvlm.py:50-70): forward all_gather + cat, backward keeps the LOCAL row slice only."""

    @staticmethod
    def forward(ctx, t):
        import torch.distributed as dist
        W, r = dist.get_world_size(), dist.get_rank()
        out = [torch.empty_like(t) for _ in range(W)]
        dist.all_gather(out, t.contiguous())
        ctx.r, ctx.b = r, t.shape[0]
        return torch.cat(out, 0)

    @staticmethod
    def backward(ctx, g):
        return g[ctx.b * ctx.r: ctx.b * (ctx.r + 1)]


def gather_rows(t):
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return t
    return _SliceBackwardGather.apply(t)


# --------------------------------------------------------------------------------------- losses
def soft_labels(idx_all):
    """xvlm.py:283-287: labels = eq(idx, idx^T) / rowsum."""
    pos = torch.eq(idx_all.view(-1, 1), idx_all.view(1, -1)).float()
    return pos / pos.sum(1, keepdim=True)


def contrastive_loss(feat_a, feat_b, temp, idx=None):
    """XVLMBase.get_contrastive_loss (models/xvlm.py:260-292)."""
    a, b = gather_rows(feat_a), gather_rows(feat_b)
    logits = a @ b.t() / temp
    if idx is None:
        labels = torch.arange(a.shape[0])
        return (F.cross_entropy(logits, labels) + F.cross_entropy(logits.t(), labels)) / 2
    lab = soft_labels(gather_rows(idx.view(-1, 1)))
    i2t = -(torch.log_softmax(logits, 1) * lab).sum(1).mean()
    t2i = -(torch.log_softmax(logits.t(), 1) * lab).sum(1).mean()
    return (i2t + t2i) / 2


def caption_contrastive_loss(cq, text_feat, temp):
    """get_caption_contrastive_loss (model_retrieval_caption.py:145-152).  cq [b,n,D] batch-major; LOCAL batch."""
    sim = torch.einsum('ind,jd->nij', cq, text_feat)
    logits = sim.max(0)[0] / temp
    labels = torch.arange(text_feat.shape[0])
    return (F.cross_entropy(logits, labels) + F.cross_entropy(logits.t(), labels)) / 2


def _norm_score(s):
    """norm_score (model_retrieval_caption.py:87-90): GLOBAL min / max."""
    s = s - s.min()
    return s / s.max()


def dstl_loss(feat_v, cq, feat_s, feat_t, alpha=0.8):
    """dstl_loss (model_retrieval_caption.py:94-116).  cq [b,n,D] batch-major.  logits_tv has NO temperature."""
    v, c, s, t = gather_rows(feat_v), gather_rows(cq), gather_rows(feat_s), gather_rows(feat_t)
    logits_tv = t @ v.t()
    logits_sv = _norm_score(s @ v.t())
    # reference: sim[n*G + i, j] = <c[i,n], s[j]>, max over n -> [i (caption), j (source text)]
    logits_sc = _norm_score(torch.einsum('ind,jd->nij', c, s).max(0)[0])
    labels = torch.softmax(alpha * logits_sv + (1.0 - alpha) * logits_sc, 1).detach()
    logp = torch.log_softmax(logits_tv, 1)
    # F.kl_div(logp, labels, 'batchmean') = sum(labels * (log labels - logp)) / G
    return (torch.xlogy(labels, labels) - labels * logp).sum() / logits_tv.shape[0]


def caption_vision_loss(sd, ori_caption, ori_image, idx):
    """caption_vision_loss (model_retrieval_caption.py:118-143), exactly as written: all-gather the token
    tensors, project, F.normalize with its default dim=1 (the TOKEN axis), all token pairs, mean over both
    token axes, soft-label CE in one direction.  ori_caption [b,n,vw], ori_image [b,N,vw] batch-major."""
    img, cap = gather_rows(ori_image), gather_rows(ori_caption)
    cap = F.normalize(linear(cap, sd['cproj.weight'], sd['cproj.bias']), dim=1)
    img = F.normalize(linear(img, sd['vproj.weight'], sd['vproj.bias']), dim=1)
    sim = torch.einsum('icd,jvd->ijcv', cap, img).mean(-1).mean(-1)
    lab = soft_labels(gather_rows(idx.view(-1, 1)))
    return -(torch.log_softmax(sim, 1) * lab).sum(1).mean()


def caption_regularization(ori_caption):
    """caption_regularization (model_retrieval_caption.py:65-73).  ori_caption [b,n,vw] batch-major."""
    c = F.normalize(ori_caption, dim=-1)
    sim = c @ c.transpose(1, 2)
    return (sim - torch.eye(c.shape[1])).mean()


def get_features(sd, image_embeds=None, text_embeds=None):
    """XVLMBase.get_features (models/xvlm.py:241-256): CLS -> Linear -> L2 normalise."""
    if image_embeds is not None:
        return F.normalize(linear(image_embeds[:, 0, :], sd['vision_proj.weight'], sd['vision_proj.bias'], keep_f32=True), dim=-1)
    return F.normalize(linear(text_embeds[:, 0, :], sd['text_proj.weight'], sd['text_proj.bias'], keep_f32=True), dim=-1)


# --------------------------------------------------------------------------------------- full step
def retrieval_forward(sd, cfg, image, ids_s, mask_s, ids_t, mask_t, ids_c, mask_c, idx, text_heads):
    """RetrievalModel.forward (models/model_retrieval_caption.py:154-199) -> the 5 weighted losses."""
    vis = encode_image(sd, 'clip_encoder.visual.', image)
    with torch.no_grad():
        if cfg.get('caption_encoder_name', 'mbert') == 'clip':      # ids_c = CLIP token ids, padding id 0 (:158-167); mask_c unused
            cap_hidden = encode_text_hidden(sd, 'clip_encoder.', ids_c)
            mask_c = ids_c != 0
        else:
            cap_hidden = bert_forward(sd, 'text_encoder.', ids_c, mask_c, text_heads)
    after_image, after_caption, ori_caption = interaction_with_caption(sd, vis, cap_hidden, ~mask_c.bool())
    loss_cv = caption_vision_loss(sd, ori_caption, vis, idx) * cfg['weight_cv_loss']
    loss_reg = caption_regularization(ori_caption)
    feat_v = get_features(sd, image_embeds=after_image)
    feat_s = get_features(sd, text_embeds=bert_forward(sd, 'text_encoder.', ids_s, mask_s, text_heads))
    feat_t = get_features(sd, text_embeds=bert_forward(sd, 'text_encoder.', ids_t, mask_t, text_heads))
    temp = sd['temp']
    l_vs = contrastive_loss(feat_v, feat_s, temp, idx)
    l_vt = contrastive_loss(feat_v, feat_t, temp, idx)
    l_st = contrastive_loss(feat_s, feat_t, temp, idx)
    cq = linear(after_caption, sd['caption_proj1.weight'], sd['caption_proj1.bias'], keep_f32=True)
    l_c = caption_contrastive_loss(cq, feat_s, temp) + caption_contrastive_loss(cq, feat_t, temp)
    wd = cfg['weight_dstl_loss']
    l_vt = l_vt * (1 - wd) + dstl_loss(feat_v, cq, feat_s, feat_t) * wd
    return l_vs + loss_cv, l_vt, l_st, l_c * cfg['weight_caption_loss'], loss_reg * cfg['weight_reg_loss']


# --------------------------------------------------------------------------------------- video model
def video_encode(sd, video, video_mask, heads=8):
    """RetrievalModel.get_vision_embeds of the video model (models/video_model_retrieval_caption.py:30-34):
    SelfAttention stack (models/attention.py:87-128 - the CrossAttentionLayer arithmetic with q = k = v = src) over the
    frame features, padded frames masked as keys.  video [b,F,vw], video_mask [b,F] bool (True = real frame).
    Returns (frame tokens [b,F,vw], key padding mask = ~video_mask)."""
    kpm = ~video_mask.bool()
    out, i = video, 0
    while f'vid_encoder.layers.{i}.linear1.weight' in sd:          # each layer attends to ITS OWN input (q = k = v = src)
        layer = {k: v for k, v in sd.items() if k.startswith(f'vid_encoder.layers.{i}.')}
        one = {k.replace(f'layers.{i}.', 'layers.0.', 1): v for k, v in layer.items()}
        out = cross_attention(one, 'vid_encoder.', out, out, kpm, heads)
        i += 1
    return out, kpm


def video_interaction_with_caption(sd, video_embeds, caption_hidden, key_padding_mask, video_kpm):
    """video_model_retrieval_caption.py:96-108: as the image model, except that crossattn2 (caption queries attending to
    the frames) masks the padded frames; crossattn (frames attending to the caption queries) is unmasked."""
    b = video_embeds.shape[0]
    queries = _r(sd['queries'].permute(1, 0, 2).expand(b, -1, -1))
    cap = linear(caption_hidden, sd['caption_proj.weight'], sd['caption_proj.bias'])
    ori_caption = cross_attention(sd, 'crossattn_query.', queries, cap, key_padding_mask)
    after_video = cross_attention(sd, 'crossattn.', video_embeds, ori_caption)
    after_caption = cross_attention(sd, 'crossattn2.', ori_caption, video_embeds, video_kpm)
    return after_video, after_caption, ori_caption


def video_get_features(sd, video_embeds, video_mask):
    """XVLMBase_video.get_features, vis_pooling='mean' (models/xvlm_video.py:260-272): masked mean over the frames ->
    vision_proj -> L2 normalise."""
    m = video_mask.unsqueeze(-1).to(video_embeds.dtype)
    pooled = _r((video_embeds * m).sum(1) / m.sum(1))
    return F.normalize(linear(pooled, sd['vision_proj.weight'], sd['vision_proj.bias'], keep_f32=True), dim=-1)


def video_caption_vision_loss(sd, ori_caption, ori_video, idx):
    """caption_vision_loss of the video model (video_model_retrieval_caption.py:144-169): the image model's arithmetic on
    the LOCAL batch (its all-gathers are commented out), padded frames included as they are."""
    cap = F.normalize(linear(ori_caption, sd['cproj.weight'], sd['cproj.bias']), dim=1)
    vid = F.normalize(linear(ori_video, sd['vproj.weight'], sd['vproj.bias']), dim=1)
    sim = torch.einsum('icd,jvd->ijcv', cap, vid).mean(-1).mean(-1)
    lab = soft_labels(idx.view(-1, 1))
    return -(torch.log_softmax(sim, 1) * lab).sum(1).mean()


def video_retrieval_forward(sd, cfg, video, video_mask, ids_s, mask_s, ids_t, mask_t, ids_c, mask_c, idx, text_heads):
    """RetrievalModel.forward of the video model (video_model_retrieval_caption.py:180-227) -> the 5 weighted losses."""
    vid, vkpm = video_encode(sd, video, video_mask)
    with torch.no_grad():
        cap_hidden = bert_forward(sd, 'text_encoder.', ids_c, mask_c, text_heads)
    after_video, after_caption, ori_caption = video_interaction_with_caption(sd, vid, cap_hidden, ~mask_c.bool(), vkpm)
    loss_cv = video_caption_vision_loss(sd, ori_caption, vid, idx) * cfg['weight_cv_loss']
    loss_reg = caption_regularization(ori_caption)
    feat_v = video_get_features(sd, after_video, video_mask)
    feat_s = get_features(sd, text_embeds=bert_forward(sd, 'text_encoder.', ids_s, mask_s, text_heads))
    feat_t = get_features(sd, text_embeds=bert_forward(sd, 'text_encoder.', ids_t, mask_t, text_heads))
    temp = sd['temp']
    l_vs = contrastive_loss(feat_v, feat_s, temp, idx)
    l_vt = contrastive_loss(feat_v, feat_t, temp, idx)
    l_st = contrastive_loss(feat_s, feat_t, temp, idx)
    cq = linear(after_caption, sd['caption_proj1.weight'], sd['caption_proj1.bias'], keep_f32=True)
    l_c = caption_contrastive_loss(cq, feat_s, temp) + caption_contrastive_loss(cq, feat_t, temp)
    wd = cfg['weight_dstl_loss']
    l_vt = l_vt * (1 - wd) + dstl_loss(feat_v, cq, feat_s, feat_t) * wd
    return l_vs + loss_cv, l_vt, l_st, l_c * cfg['weight_caption_loss'], loss_reg * cfg['weight_reg_loss']


# --------------------------------------------------------------------------------------- evaluation
def itm_eval(scores_i2t, scores_t2i, txt2img, img2txt):
    """itm_eval (image_Retrieval_caption.py:261-317).  rank of a ground-truth item = number of strictly larger
    scores in its row (== position in the descending argsort for tie-free scores)."""
    scores_i2t, scores_t2i = np.asarray(scores_i2t), np.asarray(scores_t2i)
    ranks = np.zeros(scores_i2t.shape[0])
    for i, row in enumerate(scores_i2t):
        ranks[i] = min(int((row > row[t]).sum()) for t in img2txt[i])
    tr = [100.0 * float((ranks < k).sum()) / len(ranks) for k in (1, 5, 10)]
    ranks = np.zeros(scores_t2i.shape[0])
    for t, row in enumerate(scores_t2i):
        ranks[t] = int((row > row[txt2img[t]]).sum())
    ir = [100.0 * float((ranks < k).sum()) / len(ranks) for k in (1, 5, 10)]
    tr_mean, ir_mean = sum(tr) / 3, sum(ir) / 3
    return {'txt_r1': tr[0], 'txt_r5': tr[1], 'txt_r10': tr[2], 'txt_r_mean': tr_mean, 'txt_sum_r': sum(tr),
            'img_r1': ir[0], 'img_r5': ir[1], 'img_r10': ir[2], 'img_r_mean': ir_mean,
            'r_mean': (tr_mean + ir_mean) / 2, 'img_sumr': sum(ir),
            'sumr_avg': np.round((sum(tr) + sum(ir)) / 6, 2), 'sumr_sum': sum(tr) + sum(ir)}
