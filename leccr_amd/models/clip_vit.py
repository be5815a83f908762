"""CLIP vision tower with the reference's module tree (clip/model.py:171-240, 246-346) over the HIP ops.

The ViT image path is live in every configuration.  The CLIP text tower (clip_encoder.transformer.*, token_embedding, ...) receives no
gradient in any configuration of the reference and stays outside the parameter arena; with caption_encoder_name = 'clip'
(models/model_retrieval_caption.py:57-58) it runs forward-only, under no_grad, as the caption encoder: encode_text below, on static compute
copies made once (prepare_text_tower).
"""
import numpy as np
import torch
import torch.nn as nn

from .. import ops
from .layers import LayerNormParams, Linear, mark_unused


class _MHAParams(nn.Module):
    """nn.MultiheadAttention's parameter names: in_proj_weight [3d,d], in_proj_bias, out_proj.{weight,bias}."""

    def __init__(self, d, xavier=True):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * d, d))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * d))
        self.out_proj = Linear(d, d)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.zeros_(self.out_proj.bias)


class _MLP(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.c_fc = Linear(d, 4 * d)
        self.c_proj = Linear(4 * d, d)


class ResidualAttentionBlock(nn.Module):
    def __init__(self, d, heads):
        super().__init__()
        self.heads = heads
        self.fp8 = False               # set by XVLMBase._ensure_runtime for compute_dtype 'fp8'
        self.attn = _MHAParams(d)
        self.ln_1 = LayerNormParams(d, 1e-5)
        self.mlp = _MLP(d)
        self.ln_2 = LayerNormParams(d, 1e-5)

    def _params(self):
        return (self.ln_1.weight, self.ln_1.bias, self.attn.in_proj_weight, self.attn.in_proj_bias,
                self.attn.out_proj.weight, self.attn.out_proj.bias, self.ln_2.weight, self.ln_2.bias,
                self.mlp.c_fc.weight, self.mlp.c_fc.bias, self.mlp.c_proj.weight, self.mlp.c_proj.bias)

    def forward(self, x):
        return ops.ViTBlockFn.apply(x, self, *self._params())


class Transformer(nn.Module):
    def __init__(self, width, layers, heads):
        super().__init__()
        self.width, self.layers = width, layers
        self.resblocks = nn.Sequential(*[ResidualAttentionBlock(width, heads) for _ in range(layers)])

    def forward(self, x):
        return self.resblocks(x)


class VisionTransformer(nn.Module):
    def __init__(self, input_resolution, patch_size, width, layers, heads, output_dim):
        super().__init__()
        self.input_resolution, self.patch_size, self.output_dim = input_resolution, patch_size, output_dim
        self.conv1 = nn.Conv2d(3, width, patch_size, patch_size, bias=False)      # parameter holder (weight [d,3,p,p])
        scale = width ** -0.5
        self.class_embedding = nn.Parameter(scale * torch.randn(width))
        self.positional_embedding = nn.Parameter(scale * torch.randn((input_resolution // patch_size) ** 2 + 1, width))
        self.ln_pre = LayerNormParams(width, 1e-5)
        self.transformer = Transformer(width, layers, heads)
        self.ln_post = LayerNormParams(width, 1e-5)
        self.proj = nn.Parameter(scale * torch.randn(width, output_dim))

    def forward(self, image, dtype):
        """-> transformer trunk output [b, N, width] (batch-major; clip/model.py:223-240)."""
        tok = ops.PatchEmbedFn.apply(image, self.conv1.weight, self.class_embedding, self.positional_embedding, dtype)
        x = self.ln_pre(tok)
        return self.transformer(x)


class CLIP(nn.Module):
    """Constructor signature of the reference's CLIP (clip/model.py:246-297), ViT vision towers only."""

    def __init__(self, embed_dim, image_resolution, vision_layers, vision_width, vision_patch_size, context_length,
                 vocab_size, transformer_width, transformer_heads, transformer_layers, text_tower=True):
        super().__init__()
        assert not isinstance(vision_layers, (tuple, list)), 'ModifiedResNet towers are outside the LECCR path'
        self.visual = VisionTransformer(image_resolution, vision_patch_size, vision_width, vision_layers,
                                        vision_width // 64, embed_dim)
        self.context_length = context_length
        if text_tower:      # allocated by the reference, never used by LECCR: inert parameters
            self.transformer = mark_unused(Transformer(transformer_width, transformer_layers, transformer_heads))
            self.token_embedding = mark_unused(nn.Embedding(vocab_size, transformer_width))
            self.positional_embedding = nn.Parameter(torch.empty(context_length, transformer_width))
            self.ln_final = mark_unused(LayerNormParams(transformer_width, 1e-5))
            self.text_projection = nn.Parameter(torch.empty(transformer_width, embed_dim))
            self.logit_scale = nn.Parameter(torch.ones([]) * np.log(1 / 0.07))
            for p in (self.positional_embedding, self.text_projection, self.logit_scale):
                p._leccr_unused = True
            nn.init.normal_(self.token_embedding.weight, std=0.02)
            nn.init.normal_(self.positional_embedding, std=0.01)
            nn.init.normal_(self.text_projection, std=transformer_width ** -0.5)
            self._init_blocks(self.transformer)
        self._init_blocks(self.visual.transformer)
        self.compute_dtype = torch.float32

    @staticmethod
    def _init_blocks(tr):
        """initialize_parameters (clip/model.py:299-326)."""
        proj_std = (tr.width ** -0.5) * ((2 * tr.layers) ** -0.5)
        attn_std = tr.width ** -0.5
        fc_std = (2 * tr.width) ** -0.5
        for blk in tr.resblocks:
            nn.init.normal_(blk.attn.in_proj_weight, std=attn_std)
            nn.init.normal_(blk.attn.out_proj.weight, std=proj_std)
            nn.init.normal_(blk.mlp.c_fc.weight, std=fc_std)
            nn.init.normal_(blk.mlp.c_proj.weight, std=proj_std)

    def float(self):
        return self

    # ------------------------------------------------------------------ text tower (caption_encoder_name = 'clip')
    def prepare_text_tower(self, dtype):
        """Compute copies of the text tower's matrices in `dtype` (they are never trained - the reference encodes captions under
        no_grad and nothing else reaches these parameters - so the copies are made once), causal blocks."""
        for blk in self.transformer.resblocks:
            blk.causal, blk.fp8 = True, False
            for w in (blk.attn.in_proj_weight, blk.attn.out_proj.weight, blk.mlp.c_fc.weight, blk.mlp.c_proj.weight):
                w._lp, w._lpt = w.data.to(dtype).contiguous(), None
        tp = self.text_projection
        tp._lp, tp._lpt = tp.data.to(dtype).contiguous(), tp.data.t().to(dtype).contiguous()
        self._text_zero_type = torch.zeros(self.token_embedding.weight.shape[1], device=tp.device, dtype=torch.float32)
        self._text_ready = True

    @torch.no_grad()
    def encode_text(self, text, return_hidden=False):
        """clip/model.py:348-365: token + positional embedding -> causal transformer -> ln_final on ALL tokens @ text_projection;
        x = the row of each sequence's highest token id (the EOT token).  Forward only."""
        assert getattr(self, '_text_ready', False), 'the CLIP text tower is live for caption_encoder_name = "clip" only'
        assert text.dim() == 2 and text.shape[1] == self.positional_embedding.shape[0], 'captions must be tokenised to context_length'
        from .. import kernels as K
        x = K.bert_embed_fwd(text.contiguous(), self.token_embedding.weight.data, self.positional_embedding.data, self._text_zero_type,
                             self.compute_dtype)                                   # word[ids] + position (the type row is zero)
        x = self.transformer(x)
        hidden = ops.LnProjFn.apply(x, self.ln_final.weight, self.ln_final.bias, self.text_projection)
        x = hidden[torch.arange(hidden.shape[0], device=hidden.device), text.argmax(dim=-1)]
        if return_hidden:
            return x, hidden
        return x

    def encode_image(self, image, return_hidden=False):
        """clip/model.py:340-346: ln_post on ALL tokens then @ proj; returns (cls, hidden) when return_hidden."""
        v = self.visual
        x = v(image, self.compute_dtype)
        hidden = ops.LnProjFn.apply(x, v.ln_post.weight, v.ln_post.bias, v.proj)
        cls = hidden[:, 0, :]
        if return_hidden:
            return cls.contiguous(), hidden
        return cls.contiguous()


# name -> (width, layers, patch, embed_dim) for the ViT entries of clip.clip._MODELS (clip/clip.py:30-40)
VIT_GEOMETRY = {'ViT-B/32': (768, 12, 32, 512), 'ViT-B/16': (768, 12, 16, 512), 'ViT-L/14': (1024, 24, 14, 768)}


def build_clip(name='ViT-B/32', image_resolution=384, text_tower=True, geometry=None):
    """Random-init CLIP of the named geometry at the reference's hard-coded 384 resolution (clip/model.py:414).
    `geometry` (dict of CLIP constructor arguments) overrides the name - used by the tiny parity fixtures."""
    if geometry is not None:
        return CLIP(text_tower=text_tower, **geometry)
    width, layers, patch, embed = VIT_GEOMETRY[name]
    return CLIP(embed, image_resolution, layers, width, patch, 77, 49408, 512, 8, 12, text_tower=text_tower)


# --------------------------------------------------------------------------------------- raw CLIP checkpoints
def _cubic_weights(t, a=-0.75):
    """The 4 taps of the cubic-convolution kernel (Keys, a = -0.75: what torch's mode='bicubic' uses) at fraction t."""
    t = np.asarray(t, dtype=np.float64)
    w0 = ((a * (t + 1) - 5 * a) * (t + 1) + 8 * a) * (t + 1) - 4 * a
    w1 = ((a + 2) * t - (a + 3)) * t * t + 1
    w2 = ((a + 2) * (1 - t) - (a + 3)) * (1 - t) * (1 - t) + 1
    w3 = ((a * (2 - t) - 5 * a) * (2 - t) + 8 * a) * (2 - t) - 4 * a
    return np.stack([w0, w1, w2, w3], -1)


def _resize_matrix(n_in, n_out):
    """[n_out, n_in] bicubic resampling matrix, align_corners=False, border taps clamped (F.interpolate semantics)."""
    src = (np.arange(n_out, dtype=np.float64) + 0.5) * (n_in / n_out) - 0.5
    i0 = np.floor(src).astype(np.int64)
    w = _cubic_weights(src - i0)
    m = np.zeros((n_out, n_in), dtype=np.float64)
    for k in range(4):
        np.add.at(m, (np.arange(n_out), np.clip(i0 - 1 + k, 0, n_in - 1)), w[:, k])
    return m


def interpolate_pos_embed(pos_embed_checkpoint, num_patches, num_extra_tokens=1):
    """clip/model.py:456-481: bicubically resize the [1, 1+g0*g0, d] positional embedding of a CLIP checkpoint to
    [1, 1+g*g, d] (g*g = num_patches); the class-token row is kept.  Load-time host arithmetic (separable resampling as two
    small matrix products in f64), not part of the step."""
    pe = pos_embed_checkpoint
    d = pe.shape[-1]
    orig = int((pe.shape[-2] - num_extra_tokens) ** 0.5)
    new = int(num_patches ** 0.5)
    if orig == new:
        return pe
    extra = pe[:, :num_extra_tokens]
    grid = pe[:, num_extra_tokens:].reshape(-1, orig, orig, d).detach().cpu().double().numpy()
    m = _resize_matrix(orig, new)
    out = np.einsum('yi,bixd->byxd', m, grid)                             # rows
    out = np.einsum('xj,byjd->byxd', m, out)                             # columns
    out = torch.from_numpy(out.reshape(-1, new * new, d)).to(pe.dtype)
    print('reshape position embedding from %d to %d' % (orig ** 2, new ** 2))
    return torch.cat((extra, out.to(extra.device)), dim=1)


def build_model(state_dict, text_tower=True):
    """clip/model.py:408-453: a CLIP whose geometry is inferred from a raw (OpenAI-format) ViT state dict, image
    resolution hard-coded to 384 with the positional embedding resized to it, weights loaded (as f32 master weights)."""
    assert 'visual.proj' in state_dict, 'ModifiedResNet towers are outside the LECCR path'
    sd = dict(state_dict)
    image_resolution = 384
    patch = sd['visual.conv1.weight'].shape[-1]
    num_patches = (image_resolution // patch) ** 2
    sd['visual.positional_embedding'] = interpolate_pos_embed(sd['visual.positional_embedding'].unsqueeze(0).float(),
                                                              num_patches=num_patches).squeeze(0)
    width = sd['visual.conv1.weight'].shape[0]
    layers = len([k for k in sd if k.startswith('visual.') and k.endswith('.attn.in_proj_weight')])
    embed_dim = sd['text_projection'].shape[1] if 'text_projection' in sd else sd['visual.proj'].shape[1]
    has_text = text_tower and 'token_embedding.weight' in sd
    if has_text:
        tw = sd['ln_final.weight'].shape[0]
        model = CLIP(embed_dim, image_resolution, layers, width, patch, sd['positional_embedding'].shape[0],
                     sd['token_embedding.weight'].shape[0], tw, tw // 64,
                     len(set(k.split('.')[2] for k in sd if k.startswith('transformer.resblocks'))))
    else:
        model = CLIP(embed_dim, image_resolution, layers, width, patch, 77, 49408, 512, 8, 12, text_tower=False)
        sd = {k: v for k, v in sd.items() if k.startswith('visual.')}
    for key in ('input_resolution', 'context_length', 'vocab_size'):
        sd.pop(key, None)
    model.load_state_dict({k: v.float() for k, v in sd.items()})
    return model.eval()
