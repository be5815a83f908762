"""XVLMBase + AllGather with the reference's API (models/xvlm.py:50-70,150-292), computed by the HIP ops.

Extensions (all optional, defaults reproduce the reference's hard-coded choices):
  config['clip_name']      'ViT-B/32' (reference default, xvlm.py:85) | 'ViT-B/16' | 'ViT-L/14'
  config['text_config']    dict of BERT geometry overrides (default: bert-base-multilingual-cased geometry)
  config['clip_geometry']  dict of CLIP constructor arguments overriding clip_name (tiny parity fixtures)
  config['compute_dtype']  'fp32' (DEFAULT = what the reference's configs get: reference precision, f32 activations, every large product on
                           the matrix cores as bf16 hi / lo splits - leccr_split3 - and the f32 MFMA flash attention; LECCR_F32_MFMA=0: the
                           scalar f32 kernels) | 'bf16' (MFMA path, BASELINE config B) | 'fp8' (BASELINE configs[4]: the bf16 path with
                           OCP fp8 operands - e4m3 forward, e5m2 gradients, per-tensor delayed scaling - in the forward and dX GEMMs of
                           the ViT blocks and BERT layers; LayerNorm, softmax, residuals, weight gradients, heads and losses as in bf16)
  config['clip_text_tower'] keep the unused CLIP text tower's parameters for state_dict compatibility (default True)
"""
import os

import torch
import torch.distributed as dist
import torch.nn as nn

from .. import kernels, ops
from ..arena import ParamArena
from ..parallel import collective_path
from .bert import BertModel
from .clip_vit import VIT_GEOMETRY, build_clip
from .layers import Linear, mark_unused


class AllGather(torch.autograd.Function):
    """models/xvlm.py:50-70: forward all_gather + cat(dim 0); backward returns ONLY the local row slice of the
    incoming gradient (no reduce-scatter).  Over RCCL (backend 'nccl' on ROCm) / xGMI."""

    @staticmethod
    def forward(ctx, tensor, rank, world_size):
        tensor = tensor.contiguous()
        out = torch.empty((world_size,) + tuple(tensor.shape), device=tensor.device, dtype=tensor.dtype)
        if dist.get_backend() == 'nccl':         # RCCL: one flat all-gather straight into the output
            dist.all_gather_into_tensor(out.view(-1), tensor.view(-1))
        else:                                    # gloo (CPU tests, single-GPU rehearsal)
            dist.all_gather(list(out.unbind(0)), tensor)
        ctx.rank, ctx.batch_size = rank, tensor.shape[0]
        return out.view((world_size * tensor.shape[0],) + tuple(tensor.shape[1:]))

    @staticmethod
    def backward(ctx, grad_output):
        return grad_output[ctx.batch_size * ctx.rank: ctx.batch_size * (ctx.rank + 1)], None, None


allgather = AllGather.apply


def _dist_info():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


class AllGatherPacked(torch.autograd.Function):
    """Several f32 row blocks [b, ...] in ONE all-gather (their flat concatenation), same backward as AllGather for each."""

    @staticmethod
    def forward(ctx, rank, world_size, *tensors):
        flat = torch.cat([t.reshape(-1) for t in tensors])
        out = torch.empty(world_size, flat.numel(), device=flat.device, dtype=flat.dtype)
        if dist.get_backend() == 'nccl':
            dist.all_gather_into_tensor(out.view(-1), flat)
        else:
            dist.all_gather(list(out.unbind(0)), flat)
        ctx.rank, ctx.rows = rank, [t.shape[0] for t in tensors]
        res, off = [], 0
        for t in tensors:
            n = t.numel()
            res.append(out[:, off:off + n].reshape((world_size * t.shape[0],) + tuple(t.shape[1:])))
            off += n
        return tuple(res)

    @staticmethod
    def backward(ctx, *grads):
        r = ctx.rank
        return (None, None) + tuple(None if g is None else g[b * r: b * (r + 1)] for g, b in zip(grads, ctx.rows))


class StepGathers:
    """One training step gathers the same feature blocks for several loss terms (image_feat for three of them, idx for four): the
    gathered copy is kept per source tensor for the duration of the step (begin() clears it), and packed() fills it for the whole
    set with a single collective.  One instance per MODEL (XVLMBase.gathers): two models stepping in one process (a teacher beside
    a student, a test's oracle twin) never see each other's blocks - the cache key is an address, and addresses are only unique
    among the tensors one instance keeps alive."""

    def __init__(self):
        self.hit = {}

    @staticmethod
    def _key(t):
        return (t.data_ptr(), tuple(t.shape), tuple(t.stride()), t.dtype)

    def begin(self):
        self.hit.clear()

    def rows(self, t):
        rank, world = _dist_info()
        if not collective_path(world):
            return t
        hit = self.hit.get(self._key(t))
        if hit is not None:
            return hit[1]
        out = allgather(t, rank, world)
        self.hit[self._key(t)] = (t, out)          # the source stays referenced: its address cannot be reused within the step
        return out

    def packed(self, tensors, idx=None):
        """All-gather the f32 blocks `tensors` (and the int64 column `idx`, bit-cast to two f32 words per row) in one collective
        and remember the results for rows()."""
        rank, world = _dist_info()
        if not collective_path(world):
            return
        srcs = [t for t in tensors if self._key(t) not in self.hit]
        assert all(t.dtype == torch.float32 and t.is_contiguous() for t in srcs)
        packed = list(srcs)
        if idx is not None and self._key(idx) not in self.hit:
            assert idx.dtype == torch.int64 and idx.is_contiguous() and idx.dim() == 2 and idx.shape[1] == 1
            packed.append(idx.view(torch.float32))                     # [b, 2] words, copied bit for bit
        else:
            idx = None
        if not packed:
            return
        outs = AllGatherPacked.apply(rank, world, *packed)
        for t, o in zip(srcs, outs):
            self.hit[self._key(t)] = (t, o)
        if idx is not None:
            self.hit[self._key(idx)] = (idx, outs[-1].detach().contiguous().view(torch.int64))


# module-level instance + functions: the free-standing API (tests, scripts without a model)
_default_gathers = StepGathers()
_gathered = _default_gathers.hit


def begin_step_gathers():
    _default_gathers.begin()


def gather_rows(t):
    return _default_gathers.rows(t)


def gather_rows_packed(tensors, idx=None):
    _default_gathers.packed(tensors, idx)


def local_rows(b):
    """(row0, nrows) of this rank's b rows inside an all-gathered [W*b, ...] tensor, or None on a single rank."""
    rank, world = _dist_info()
    return None if not collective_path(world) else (rank * b, b)


def build_mlp(input_dim, output_dim):
    """xvlm.py:74-80 (itm_head: allocated by the reference, never used on the LECCR path)."""
    return nn.Sequential(nn.Linear(input_dim, input_dim * 2), nn.LayerNorm(input_dim * 2), nn.GELU(),
                         nn.Linear(input_dim * 2, output_dim))


def build_clip_encoder(config):
    """xvlm.py:83-88 hard-codes 'ViT-B/32' + pretrained weights; here: any ViT geometry, seeded random init
    (no network on this pipeline), overridable through config['clip_name']."""
    config = config or {}
    name = config.get('clip_name', 'ViT-B/32')
    geometry = config.get('clip_geometry')
    clip = build_clip(name, image_resolution=config.get('clip_image_res', 384),
                      text_tower=config.get('clip_text_tower', True), geometry=geometry)
    return clip, (geometry['embed_dim'] if geometry else VIT_GEOMETRY[name][3])


def build_text_encoder(config):
    """xvlm.py:91-103: bert-base-multilingual-cased geometry (random init here), dropout 0.1 in the reference."""
    enc = BertModel((config or {}).get('text_config'))
    return enc, enc.config.hidden_size


_UNUSED_MARKERS = ('.norm1.', 'itm_head.', '.pooler.')


def is_unused_parameter(name):
    """Parameters the reference allocates but that never receive a gradient."""
    if name.startswith('clip_encoder.') and not name.startswith('clip_encoder.visual.'):
        return True
    return any(m in name for m in _UNUSED_MARKERS)


class XVLMBase(nn.Module):
    def __init__(self, config=None, load_vision_params=False, load_text_params=False, use_contrastive_loss=False,
                 use_matching_loss=False, use_mlm_loss=False, use_bbox_loss=False, config_text=None):
        super().__init__()
        self.allgather = allgather
        self.init_params = []
        self.config = config
        if getattr(self, '_with_clip_encoder', True):
            self.clip_encoder, vision_width = build_clip_encoder(config)
        else:                       # XVLMBase_video (models/xvlm_video.py:173-192): frame features in, no image tower
            self.clip_encoder, vision_width = None, config['vision_width']
        self.text_encoder, text_width = build_text_encoder(config)
        self.vision_width, self.text_width = vision_width, text_width
        if use_contrastive_loss:
            self.embed_dim = config['embed_dim']
            self.text_proj = Linear(self.text_width, self.embed_dim)
            self.init_params.extend(['text_proj.' + n for n, _ in self.text_proj.named_parameters()])
            if config['use_one_cl_proj_only']:
                assert self.vision_width == self.text_width
                self.vision_proj = None
            else:
                self.vision_proj = Linear(self.vision_width, self.embed_dim)
                self.init_params.extend(['vision_proj.' + n for n, _ in self.vision_proj.named_parameters()])
            self.temp = nn.Parameter(torch.ones([]) * config['temp'])
            self.init_params.extend(['temp'])
        if use_matching_loss:
            self.itm_head = mark_unused(build_mlp(input_dim=self.text_width, output_dim=2))
            self.init_params.extend(['itm_head.' + n for n, _ in self.itm_head.named_parameters()])
        # the ViT trunk is first in forward, hence last in backward: its parameters go to the end of the arena so that
        # everything before them can be all-reduced while the ViT backward still runs (leccr_amd.parallel)
        self._arena_late_prefix = 'clip_encoder.visual.'
        dt = (config or {}).get('compute_dtype', 'fp32')
        self.compute_dtype = {'fp32': torch.float32, 'bf16': torch.bfloat16, 'fp8': torch.bfloat16}[dt]
        self.fp8 = dt == 'fp8'
        self.fp8_sites = None
        self.arena = None
        self.gathers = StepGathers()        # this model's per-step cache of all-gathered blocks

    # ------------------------------------------------------------------ runtime (arena + compute copies)
    def _ensure_runtime(self):
        """Build the ParamArena on first use (after init_caption_encoder() and .to(device), as the reference's
        main() orders them - image_Retrieval_caption.py:370-381)."""
        if self.arena is not None:
            return self.arena
        dev = next(self.parameters()).device
        if dev.type != 'cuda':
            raise RuntimeError('leccr_amd models run on an MI355X only: move the model to the GPU first '
                               '(there is no CPU fallback)')
        for n, p in self.named_parameters():
            p._leccr_unused = is_unused_parameter(n)
        self.arena = ParamArena(self, self.compute_dtype, dev)
        self.text_encoder.compute_dtype = self.compute_dtype
        a = self.arena
        vis_proj = None
        if self.clip_encoder is not None:
            self.clip_encoder.compute_dtype = self.compute_dtype
            vis = self.clip_encoder.visual
            cw = vis.conv1.weight
            cw._lp2d, cw._g2d = cw._lp.view(cw.shape[0], -1), cw._g.view(cw.shape[0], -1)
            k0 = cw._lp2d.shape[1]
            if k0 % 64 and self.compute_dtype == torch.bfloat16:
                # ViT-L/14: 3*14*14 = 588 patch columns.  The MFMA GEMMs want K (forward) / N (weight gradient) in
                # multiples of 64 / 8: a zero-padded [d, 640] bf16 copy of the weight, refreshed with the other compute copies
                kp = (k0 + 63) // 64 * 64
                cw._lp_pad = torch.zeros(cw.shape[0], kp, device=dev, dtype=torch.bfloat16)
                a.post_refresh.append(lambda cw=cw, k0=k0: cw._lp_pad[:, :k0].copy_(cw._lp2d))
            vis_proj = vis.proj
            a.need_transposed(vis_proj)
        for name, p in self.named_parameters():
            if p.dim() == 2 and not getattr(p, '_leccr_unused', False) and 'embeddings' not in name \
                    and name != 'clip_encoder.visual.positional_embedding' and p is not vis_proj:
                a.need_transposed(p)
        for layer in self.text_encoder.encoder.layer:
            a.need_transposed(layer._qkv_w)
        if self.fp8:
            self._enable_fp8(a, dev)
        elif self.compute_dtype == torch.float32:
            # reference precision (xvlm.py:88: fp32 throughout) on the matrix cores: the encoder blocks hand bf16 hi / lo split copies
            # of their f32 activations / gradients from product to product (ops.X3; kernels.f32_mfma = False at run time: scalar kernels)
            if self.clip_encoder is not None:
                blocks = list(self.clip_encoder.visual.transformer.resblocks)
                for blk in blocks:
                    blk.fp8 = ops.X3
                for prev, blk in zip(blocks[:-1], blocks[1:]):          # block i's backward output is block i-1's c_proj dY: its LayerNorm-1 backward
                    blk._dx_consumer = (prev.mlp.c_proj.weight,)        # writes that product's G-form copy (ops._ln_bwd x3_for, ops._g8_give)
            for layer in self.text_encoder.encoder.layer:
                layer.fp8 = ops.X3
            # the weights' split copies once per optimizer step, all in one launch (round 5; they used to be split at every use: ~230
            # launches of 40 - 90 us per step).  LECCR_X3_WCACHE=0: split at every use again.
            if kernels.f32_mfma and kernels.NS == 3 and os.environ.get('LECCR_X3_WCACHE', '1') != '0':
                for name, p in self.named_parameters():
                    if p.dim() == 2 and not getattr(p, '_leccr_unused', False) and 'embeddings' not in name \
                            and name != 'clip_encoder.visual.positional_embedding' and p.shape[0] % 64 == 0 and p.shape[1] % 64 == 0:
                        a.need_split3(p)
                for layer in self.text_encoder.encoder.layer:
                    a.need_split3(layer._qkv_w)
        if getattr(self, 'caption_encoder_name', None) == 'clip' and self.clip_encoder is not None:
            self.clip_encoder.prepare_text_tower(self.compute_dtype)        # forward-only caption encoder: static compute copies
        a.refresh()
        return a

    def _enable_fp8(self, a, dev):
        """fp8 GEMM operands for the encoder blocks: every big Linear gets an e4m3 copy of its weight in both layouts ([out, in] for the
        forward product, [in, out] for dX; requantised after each optimizer step under one per-tensor site) and two activation sites
        (its input, e4m3; its incoming gradient, e5m2).  Scales are delayed by one step (kernels.Fp8Sites), primed from the data on
        first use."""
        from ..kernels import Fp8Sites
        holders = []
        if self.clip_encoder is not None:
            for blk in self.clip_encoder.visual.transformer.resblocks:
                blk.fp8 = True
                holders += [blk.attn.in_proj_weight, blk.attn.out_proj.weight, blk.mlp.c_fc.weight, blk.mlp.c_proj.weight]
        for layer in (self.text_encoder.encoder.layer if os.environ.get('LECCR_FP8_TEXT', '1') != '0' else []):
            layer.fp8 = True
            holders += [layer._qkv_w, layer.attention.output.dense.weight, layer.intermediate.dense.weight, layer.output.dense.weight]
        # activation / gradient sites: delayed scaling, all updated together at the start of a step (fp8_sites.update()).  Weight
        # sites live in a block of their own that the per-step update never touches: a weight is quantised once per optimizer step and
        # its copy is consumed until the next one, so its scale is taken from the weight itself at that moment (quantize_current)
        self.fp8_sites = sites = Fp8Sites(dev, 2 * len(holders) + 8)
        self.fp8_weight_sites = wsites = Fp8Sites(dev, len(holders) + 8)
        for h in holders:
            assert h._lpt is not None and h._lp.dim() == 2
            h._wsite, h._xsite, h._gsite = wsites.new(Fp8Sites.E4M3), sites.new(Fp8Sites.E4M3), sites.new(Fp8Sites.E5M2)
            h._q8 = torch.empty(h._lp.shape, device=dev, dtype=torch.float8_e4m3fn)
            h._q8t = torch.empty(h._lpt.shape, device=dev, dtype=torch.float8_e4m3fn)

        if self.clip_encoder is not None:      # block i's backward output dx is consumed by block i-1's c_proj dX product (ops._g8_give)
            blocks = list(self.clip_encoder.visual.transformer.resblocks)
            for prev, blk in zip(blocks[:-1], blocks[1:]):
                blk._dx_consumer = (prev.mlp.c_proj.weight,)        # a tuple: not registered as this block's parameter

        # one batched requantisation of every weight per optimizer step (three launches); weights the batch cannot take (not whole
        # 64 x 64 tiles - none at the ViT / BERT geometries) go through the per-tensor passes
        from .. import kernels as K
        batch = [h for h in holders if h._lp.shape[0] % 64 == 0 and h._lp.shape[1] % 64 == 0 and h._lp.is_contiguous()]
        if os.environ.get('LECCR_FP8_BATCH_REQUANT', '1') == '0':
            batch = []
        rest = [h for h in holders if not any(h is b for b in batch)]
        run_batch = K.fp8_requantize_weights(batch) if batch else None

        def requantise():
            if run_batch is not None:
                run_batch()
            for h in rest:
                h._wsite.quantize_current(h._lp, [(h._lp, h._q8), (h._lpt, h._q8t)])
        a.post_refresh.append(requantise)

    def refresh_weights(self):
        """Re-derive the low-precision compute copies after the master weights changed (optimizer step / load)."""
        if self.arena is not None:
            self.arena.refresh()

    def load_state_dict(self, *args, **kw):
        r = super().load_state_dict(*args, **kw)
        self.refresh_weights()
        if self.arena is not None and getattr(self, 'caption_encoder_name', None) == 'clip' and self.clip_encoder is not None:
            self.clip_encoder.prepare_text_tower(self.compute_dtype)        # (outside the arena: their copies are made here)
        return r

    # ------------------------------------------------------------------ reference API
    def get_vision_embeds(self, image, image_atts=None, idx_to_group_img=None, return_hidden=True):
        """xvlm.py:203-210 -> (image_embeds [b,N,vw], image_atts [b,N] ones)."""
        assert idx_to_group_img is None, 'grouped-image branch is outside the LECCR path'
        self._ensure_runtime()
        _, image_embeds = self.clip_encoder.encode_image(image, return_hidden=True)
        image_atts = torch.ones(image_embeds.size()[:-1], dtype=torch.long, device=image.device)
        return image_embeds, image_atts

    def get_text_embeds(self, text_ids, text_atts):
        """xvlm.py:237-238."""
        self._ensure_runtime()
        return self.text_encoder(text_ids, attention_mask=text_atts, return_dict=True).last_hidden_state

    def _feat(self, proj, cls_rows):
        return ops.l2_normalize(proj(cls_rows, out_f32=True))

    def get_features(self, image_embeds=None, text_embeds=None, vis_pooling='cls'):
        """xvlm.py:241-256: CLS -> Linear -> F.normalize (f32 features)."""
        vision_proj = self.text_proj if self.vision_proj is None else self.vision_proj
        assert vis_pooling == 'cls'
        if image_embeds is None:
            return self._feat(self.text_proj, text_embeds[:, 0, :])
        if text_embeds is None:
            return self._feat(vision_proj, image_embeds[:, 0, :])
        return self._feat(vision_proj, image_embeds[:, 0, :]), self._feat(self.text_proj, text_embeds[:, 0, :])

    def get_contrastive_loss(self, image_feat, text_feat, idx=None, tval=None):
        """xvlm.py:260-292: all-gather both feature sets (+ idx), all-pairs logits / temp, soft-label InfoNCE both
        directions.  The G x G logits are never materialised (leccr_sim_infonce_*)."""
        assert image_feat.size(-1) == self.embed_dim and text_feat.size(-1) == self.embed_dim
        gather_rows = self.gathers.rows
        a, b = gather_rows(image_feat.float()), gather_rows(text_feat.float())
        idx_all = None
        if idx is not None:
            idx = idx.view(-1, 1)
            assert idx.size(0) == image_feat.size(0)
            idx_all = gather_rows(idx).view(-1).contiguous()
        return ops.SimInfoNCEFn.apply(a, b, self.temp, idx_all, True, tval, local_rows(image_feat.size(0)))
