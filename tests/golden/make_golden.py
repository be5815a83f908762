#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the REFERENCE implementation.

Runs only in the build container (needs /root/reference, CPU only).  It imports the
reference's own classes (LECCR/clip/model.py, LECCR/models/*.py) and HF BertModel (the
third-party text tower the reference instantiates at LECCR/models/xvlm.py:91-103), feeds
them seeded inputs and seeded random weights at tiny geometries, and stores inputs,
weights, outputs and gradients as .npz.  Nothing of the reference's source text is
stored - only tensors.

    python tests/golden/make_golden.py            # writes tests/golden/*.npz, *.json

The fixtures pin oracle/ (the CPU restatement) to the reference; see tests/test_oracle_golden.py.
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REF = '/root/reference/LECCR'
OUT = os.path.dirname(os.path.abspath(__file__))


# ----------------------------------------------------------------------------- reference import
def import_reference():
    """Make the reference's model classes importable (SURVEY.md section 8c recipe).

    models/vit.py and models/swin_transformer.py need `timm` (absent); only two helper
    symbols used by the never-called load_pretrained() come from them, so empty modules with
    those names are pre-seeded.  clip/__init__ pulls torchvision (absent), so clip/model.py is
    loaded by path (it needs only torch + numpy).
    """
    if REF not in sys.path:
        sys.path.insert(0, REF)
    for name, attrs in [('models.vit', ['interpolate_pos_embed']),
                        ('models.swin_transformer', ['interpolate_relative_pos_embed'])]:
        if name not in sys.modules:
            m = types.ModuleType(name)
            for a in attrs:
                setattr(m, a, None)
            sys.modules[name] = m
    spec = importlib.util.spec_from_file_location('ref_clip_model', os.path.join(REF, 'clip/model.py'))
    clip_model = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(clip_model)
    import models.xvlm as xvlm
    return clip_model, xvlm


# standalone encoder fixtures (2 heads of 64 for the ViT = the production head size)
TINY_VIT = dict(embed_dim=64, image_resolution=64, vision_layers=2, vision_width=128, vision_patch_size=16,
                context_length=8, vocab_size=32, transformer_width=64, transformer_heads=1, transformer_layers=1)
TINY_BERT = dict(vocab_size=100, hidden_size=64, num_hidden_layers=2, num_attention_heads=2,
                 intermediate_size=128, max_position_embeddings=40, hidden_dropout_prob=0.1,
                 attention_probs_dropout_prob=0.1)
# towers inside the full-model fixtures (smaller: every weight and gradient is stored)
MODEL_VIT = dict(TINY_VIT, embed_dim=32, vision_width=64)
TINY_CFG = {'embed_dim': 16, 'temp': 0.07, 'use_one_cl_proj_only': False, 'caption_encoder_name': 'mbert',
            'num_queries': 4, 'caption_ca_layer': 2, 'caption_interaction_layer': 2, 'weight_reg_loss': 0.01,
            'weight_caption_loss': 0.01, 'weight_dstl_loss': 0.5, 'weight_cv_loss': 0.01}


def randomize_(module, seed, std=0.05):
    """Seeded, non-degenerate values for every parameter (LN gains near 1, the rest N(0,std))."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in module.named_parameters():
            if n == 'temp':
                continue
            r = torch.randn(p.shape, generator=g)
            if p.dim() == 1 and ('ln' in n.lower() or 'norm' in n.lower()) and n.endswith('weight'):
                p.copy_(1.0 + 0.1 * r)
            elif n.endswith('logit_scale'):
                continue
            else:
                p.copy_(std * r)


def make_ids(g, b, L, vocab, min_len=3):
    """[CLS]=1 first, [SEP]=2 at a per-row length, 0 padding after (SURVEY.md 8d layout, tiny vocab)."""
    ids = torch.randint(5, vocab, (b, L), generator=g)
    lens = torch.randint(min_len, L + 1, (b,), generator=g)
    lens[0] = L
    pos = torch.arange(L)[None, :]
    ids[:, 0] = 1
    for i in range(b):
        ids[i, lens[i] - 1] = 2
    mask = (pos < lens[:, None]).long()
    ids = ids * mask
    return ids, mask


class Enc:
    def __init__(self, ids, mask):
        self.input_ids, self.attention_mask = ids, mask


def sd_np(module, prefix='w/'):
    # caption_encoder.* is an alias of text_encoder.* (model_retrieval_caption.py:35): stored once
    return {prefix + k: v.detach().cpu().numpy() for k, v in module.state_dict().items()
            if not k.startswith('caption_encoder.')}


def grads_np(module, prefix='g/'):
    return {prefix + n: p.grad.detach().cpu().numpy() for n, p in module.named_parameters() if p.grad is not None}


# ----------------------------------------------------------------------------- fixtures
def fixture_vit(clip_model):
    torch.manual_seed(0)
    clip = clip_model.CLIP(**TINY_VIT).float().eval()
    randomize_(clip.visual, 11)
    g = torch.Generator().manual_seed(12)
    image = torch.randn(3, 3, 64, 64, generator=g).requires_grad_(True)
    cls, hidden = clip.encode_image(image, return_hidden=True)
    # intermediate: output of the transformer trunk before ln_post/proj
    trunk = clip.visual(image)
    probe = torch.randn(hidden.shape, generator=g)
    (hidden * probe).sum().backward()
    out = {'image': image.detach().numpy(), 'hidden': hidden.detach().numpy(), 'cls': cls.detach().numpy(),
           'trunk': trunk.detach().numpy(), 'probe': probe.numpy(), 'g_image': image.grad.numpy()}
    out.update(sd_np(clip.visual))
    out.update({k: v for k, v in grads_np(clip.visual).items() if 'resblocks.1.' not in k})
    np.savez_compressed(os.path.join(OUT, 'vit_tiny.npz'), **out)
    print('vit_tiny', hidden.shape, float(hidden.abs().mean()))


def build_bert(seed):
    from transformers import BertConfig, BertModel
    bert = BertModel(BertConfig(**TINY_BERT)).eval()
    randomize_(bert, seed)
    return bert


def fixture_bert():
    bert = build_bert(21)
    g = torch.Generator().manual_seed(22)
    ids, mask = make_ids(g, 3, 9, TINY_BERT['vocab_size'])
    h = bert(ids, attention_mask=mask, return_dict=True).last_hidden_state
    probe = torch.randn(h.shape, generator=g)
    (h * probe).sum().backward()
    out = {'ids': ids.numpy(), 'mask': mask.numpy(), 'last_hidden_state': h.detach().numpy(), 'probe': probe.numpy()}
    out.update(sd_np(bert))
    out.update(grads_np(bert))
    np.savez_compressed(os.path.join(OUT, 'bert_tiny.npz'), **out)
    print('bert_tiny', h.shape, float(h.abs().mean()))


def build_model(clip_model, xvlm, seed=31, cfg=None):
    """Reference RetrievalModel at tiny geometry with seeded weights."""
    def _clip(config=None):
        return clip_model.CLIP(**MODEL_VIT).float(), MODEL_VIT['embed_dim']

    def _bert(config=None):
        from transformers import BertConfig, BertModel
        return BertModel(BertConfig(**TINY_BERT)), TINY_BERT['hidden_size']

    xvlm.build_clip_encoder = _clip
    xvlm.build_text_encoder = _bert
    from models.model_retrieval_caption import RetrievalModel
    torch.manual_seed(seed)
    model = RetrievalModel(dict(cfg or TINY_CFG))
    model.init_caption_encoder()
    randomize_(model, seed)
    return model.eval()


def make_batch(seed, b):
    g = torch.Generator().manual_seed(seed)
    image = torch.randn(b, 3, 64, 64, generator=g)
    ids_s, m_s = make_ids(g, b, 7, TINY_BERT['vocab_size'])
    ids_t, m_t = make_ids(g, b, 9, TINY_BERT['vocab_size'])
    ids_c, m_c = make_ids(g, b, 12, TINY_BERT['vocab_size'])
    idx = torch.arange(b) + 100
    idx[b - 1] = idx[1]  # one duplicate -> soft labels (xvlm.py:283-290)
    return dict(image=image, ids_s=ids_s, mask_s=m_s, ids_t=ids_t, mask_t=m_t, ids_c=ids_c, mask_c=m_c, idx=idx)


def run_forward(model, batch, sl=slice(None)):
    return model(batch['image'][sl], [Enc(batch['ids_s'][sl], batch['mask_s'][sl]),
                                      Enc(batch['ids_t'][sl], batch['mask_t'][sl])],
                 captions=Enc(batch['ids_c'][sl], batch['mask_c'][sl]), idx=batch['idx'][sl])


def fixture_model_w1(clip_model, xvlm):
    model = build_model(clip_model, xvlm)
    batch = make_batch(32, 8)
    # intermediates through the public helper methods (model_retrieval_caption.py:154-170)
    with torch.no_grad():
        v, _ = model.get_vision_embeds(batch['image'])
        c = model.get_caption_embeds(batch['ids_c'], batch['mask_c'])
        a_img, a_cap, o_cap = model.interaction_with_caption(image_embeds=v, caption_embeds=c,
                                                             key_padding_mask=~batch['mask_c'].bool())
        feat_v = model.get_features(a_img.transpose(0, 1).contiguous(), None)
        feat_s = model.get_features(None, model.get_text_embeds(batch['ids_s'], batch['mask_s']))
        feat_t = model.get_features(None, model.get_text_embeds(batch['ids_t'], batch['mask_t']))
    losses = run_forward(model, batch)
    sum(losses).backward()
    out = {k: v_.numpy() for k, v_ in batch.items()}
    out.update({'vision_embeds': v.numpy(), 'caption_hidden': c.numpy(), 'after_image': a_img.numpy(),
                'after_caption': a_cap.numpy(), 'ori_caption': o_cap.numpy(), 'feat_v': feat_v.numpy(),
                'feat_s': feat_s.numpy(), 'feat_t': feat_t.numpy(),
                'losses': np.array([float(l) for l in losses], dtype=np.float64)})
    out.update(sd_np(model))
    out.update(grads_np(model))
    np.savez_compressed(os.path.join(OUT, 'model_tiny_w1.npz'), **out)
    print('model_tiny_w1 losses', out['losses'])


def fixture_model_clipcap(clip_model, xvlm):
    """caption_encoder_name = 'clip' (model_retrieval_caption.py:37-38, 57-58, 158-167): the captions are CLIP token ids
    [b, context_length] encoded by clip_encoder.encode_text under no_grad; key padding = (ids == 0)."""
    model = build_model(clip_model, xvlm, seed=37, cfg=dict(TINY_CFG, caption_encoder_name='clip'))
    batch = make_batch(38, 8)
    g = torch.Generator().manual_seed(39)
    L, vocab = MODEL_VIT['context_length'], MODEL_VIT['vocab_size']
    cap = torch.randint(1, vocab - 1, (8, L), generator=g)
    lens = torch.randint(3, L + 1, (8,), generator=g)
    lens[0] = L
    for i in range(8):
        cap[i, lens[i] - 1] = vocab - 1                 # the EOT token is the highest id (clip/model.py:358-359)
        cap[i, lens[i]:] = 0
    with torch.no_grad():
        c = model.get_caption_embeds(cap, None)
    losses = model(batch['image'], [Enc(batch['ids_s'], batch['mask_s']), Enc(batch['ids_t'], batch['mask_t'])],
                   captions=cap, idx=batch['idx'])
    sum(losses).backward()
    out = {k: v_.numpy() for k, v_ in batch.items() if k not in ('ids_c', 'mask_c')}
    out.update({'captions': cap.numpy(), 'caption_hidden': c.numpy(), 'losses': np.array([float(l) for l in losses], dtype=np.float64)})
    out.update(sd_np(model))
    out.update(grads_np(model))
    np.savez_compressed(os.path.join(OUT, 'model_tiny_clipcap.npz'), **out)
    print('model_tiny_clipcap losses', out['losses'])


def _w2_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(2)
    clip_model, xvlm = import_reference()
    model = build_model(clip_model, xvlm)
    ddp = torch.nn.parallel.DistributedDataParallel(model, find_unused_parameters=True)
    batch = make_batch(32, 8)
    b = 8 // world
    sl = slice(rank * b, (rank + 1) * b)
    losses = ddp(batch['image'][sl], [Enc(batch['ids_s'][sl], batch['mask_s'][sl]),
                                      Enc(batch['ids_t'][sl], batch['mask_t'][sl])],
                 captions=Enc(batch['ids_c'][sl], batch['mask_c'][sl]), idx=batch['idx'][sl])
    sum(losses).backward()
    out = {'losses': np.array([float(l) for l in losses], dtype=np.float64)}
    out.update(grads_np(model))
    ret[rank] = out
    dist.barrier()
    dist.destroy_process_group()


def fixture_model_w2():
    """Same global batch as model_tiny_w1 sharded over 2 ranks under gloo DDP (xvlm.py:50-70 semantics)."""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_w2_worker, args=(2, 29611, ret), nprocs=2, join=True)
    out = {}
    for r in (0, 1):
        for k, v in ret[r].items():
            if r == 0 or k == 'losses':      # DDP-averaged gradients are identical on both ranks
                out[f'r{r}/{k}'] = v
            else:
                assert np.allclose(v, ret[0][k], atol=1e-7), k
    np.savez_compressed(os.path.join(OUT, 'model_tiny_w2.npz'), **out)
    print('model_tiny_w2 losses', out['r0/losses'], out['r1/losses'])


def fixture_model_w4():
    """The same global batch over FOUR ranks of the reference under gloo DDP (SURVEY.md 8(c) `losses_w4`; xvlm.py:50-70,
    model_retrieval_caption.py:94-143): per-rank losses and the DDP-averaged gradients, with the 1/W-vs-full scaling at W = 4."""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_w2_worker, args=(4, 29613, ret), nprocs=4, join=True)
    out = {}
    for r in range(4):
        for k, v in ret[r].items():
            if r == 0 or k == 'losses':
                out[f'r{r}/{k}'] = v
            else:
                assert np.allclose(v, ret[0][k], atol=1e-7), k
    np.savez_compressed(os.path.join(OUT, 'model_tiny_w4.npz'), **out)
    print('model_tiny_w4 losses', [out[f'r{r}/losses'] for r in range(4)])


def fixture_losses(clip_model, xvlm):
    """Each loss function on its own, with gradients w.r.t. its inputs and temp."""
    model = build_model(clip_model, xvlm)
    D, vw, n, b, N = TINY_CFG['embed_dim'], MODEL_VIT['embed_dim'], TINY_CFG['num_queries'], 6, 5
    g = torch.Generator().manual_seed(41)
    nrm = lambda x: torch.nn.functional.normalize(x, dim=-1)
    fv = nrm(torch.randn(b, D, generator=g)).requires_grad_(True)
    fs = nrm(torch.randn(b, D, generator=g)).requires_grad_(True)
    ft = nrm(torch.randn(b, D, generator=g)).requires_grad_(True)
    cq = torch.randn(n, b, D, generator=g).requires_grad_(True)          # caption_proj1 output, LND
    cap = torch.randn(n, b, vw, generator=g).requires_grad_(True)         # ori_caption_embeds, LND
    img = torch.randn(b, N, vw, generator=g).requires_grad_(True)         # ori_image_embeds
    idx = torch.tensor([7, 3, 9, 3, 5, 7])
    out = dict(fv=fv, fs=fs, ft=ft, cq=cq, cap=cap, img=img)
    out = {k: v.detach().numpy() for k, v in out.items()}
    out['idx'] = idx.numpy()

    def rec(name, loss, inputs):
        for p in model.parameters():
            p.grad = None
        for t in inputs.values():
            t.grad = None
        loss.backward()
        out[f'{name}/loss'] = np.float64(loss.item())
        for k, t in inputs.items():  # inputs that only feed detached labels get no gradient -> zeros
            out[f'{name}/g_{k}'] = (t.grad if t.grad is not None else torch.zeros_like(t)).numpy().copy()
        for pn in ('temp', 'cproj.weight', 'cproj.bias', 'vproj.weight', 'vproj.bias'):
            p = dict(model.named_parameters())[pn]
            if p.grad is not None:
                out[f'{name}/g_{pn}'] = p.grad.numpy().copy()

    rec('itc_idx', model.get_contrastive_loss(fv, fs, idx=idx), dict(fv=fv, fs=fs))
    rec('itc_noidx', model.get_contrastive_loss(fv, ft, idx=None), dict(fv=fv, ft=ft))
    rec('cap_itc', model.get_caption_contrastive_loss(cq, fs), dict(cq=cq, fs=fs))
    rec('dstl', model.dstl_loss(fv, cq, fs, ft, idx), dict(fv=fv, cq=cq, fs=fs, ft=ft))
    rec('cv', model.caption_vision_loss(cap, img, idx), dict(cap=cap, img=img))
    rec('reg', model.caption_regularization(cap), dict(cap=cap))
    for pn in ('temp', 'cproj.weight', 'cproj.bias', 'vproj.weight', 'vproj.bias'):
        out['w/' + pn] = dict(model.named_parameters())[pn].detach().numpy()
    np.savez_compressed(os.path.join(OUT, 'losses.npz'), **out)
    print('losses', {k: float(v) for k, v in out.items() if k.endswith('/loss')})


def fixture_itm_eval():
    """itm_eval (image_Retrieval_caption.py:261-317) on tie-free scores: integer ranks pinned bit-exact."""
    if REF not in sys.path:
        sys.path.insert(0, REF)
    # the task script imports ruamel/dataset at module import; load only the function's module-level deps
    src_path = os.path.join(REF, 'image_Retrieval_caption.py')
    import ast
    tree = ast.parse(open(src_path).read())
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == 'itm_eval'][0]
    fn.decorator_list = []
    ns = {'np': np}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), src_path, 'exec'), ns)
    itm_eval = ns['itm_eval']
    rng = np.random.default_rng(51)
    cases = {}
    for name, (ni, nt, per) in {'sq': (50, 50, 1), 'five': (20, 100, 5)}.items():
        s = rng.permutation(ni * nt).reshape(ni, nt).astype(np.float32) / (ni * nt)  # tie-free
        # plant a correlated diagonal so R@K is non-trivial
        img2txt = {i: list(range(i * per, (i + 1) * per)) for i in range(ni)}
        txt2img = {t: t // per for t in range(nt)}
        for i in range(ni):
            for t in img2txt[i]:
                if rng.random() < 0.6:
                    s[i, t] += 0.7
        res = itm_eval(s, s.T.copy(), txt2img, img2txt)
        cases[name] = {'result': {k: float(v) for k, v in res.items()}, 'per': per}
        np.save(os.path.join(OUT, f'itm_scores_{name}.npy'), s)
    json.dump(cases, open(os.path.join(OUT, 'itm_eval.json'), 'w'), indent=1, sort_keys=True)
    print('itm_eval', {k: v['result']['sumr_sum'] for k, v in cases.items()})


# ----------------------------------------------------------------------------- video model (SURVEY.md 8c `video_step`)
VIDEO_CFG = dict(TINY_CFG, vision_width=64, vision_layer=2)


def import_reference_video():
    """models/video_model_retrieval_caption.py does `from models import XVLMBase_video`, which models/__init__.py has
    commented out, and xvlm_video.py imports models.box_ops (torchvision): pre-seed an empty box_ops module, import
    xvlm_video and publish its class under the name the video model asks for."""
    import_reference()
    import models
    if 'models.box_ops' not in sys.modules:
        sys.modules['models.box_ops'] = types.ModuleType('models.box_ops')
        models.box_ops = sys.modules['models.box_ops']
    import models.xvlm_video as xvlm_video
    models.XVLMBase_video = xvlm_video.XVLMBase_video
    return xvlm_video


def build_video_model(seed=41):
    xvlm_video = import_reference_video()

    def _bert(config=None):
        from transformers import BertConfig, BertModel
        return BertModel(BertConfig(**TINY_BERT)), TINY_BERT['hidden_size']

    xvlm_video.build_text_encoder = _bert
    from models.video_model_retrieval_caption import RetrievalModel
    torch.manual_seed(seed)
    model = RetrievalModel(dict(VIDEO_CFG))
    model.init_caption_encoder()
    randomize_(model, seed)
    return model.eval()


def make_video_batch(seed, b, frames=8):
    g = torch.Generator().manual_seed(seed)
    video = torch.randn(b, frames, VIDEO_CFG['vision_width'], generator=g)
    lens = torch.randint(3, frames + 1, (b,), generator=g)
    lens[0], lens[1] = frames, 3                       # one full clip, one at the minimum; the rest ragged
    vmask = torch.arange(frames)[None, :] < lens[:, None]          # True = real frame
    ids_s, m_s = make_ids(g, b, 7, TINY_BERT['vocab_size'])
    ids_t, m_t = make_ids(g, b, 9, TINY_BERT['vocab_size'])
    ids_c, m_c = make_ids(g, b, 12, TINY_BERT['vocab_size'])
    idx = torch.arange(b) + 100
    idx[b - 1] = idx[1]
    return dict(video=video, video_mask=vmask, ids_s=ids_s, mask_s=m_s, ids_t=ids_t, mask_t=m_t, ids_c=ids_c,
                mask_c=m_c, idx=idx)


def fixture_video_w1():
    model = build_video_model()
    batch = make_video_batch(52, 8)
    with torch.no_grad():
        v, vm = model.get_vision_embeds(batch['video'], batch['video_mask'])
        c = model.get_caption_embeds(batch['ids_c'], batch['mask_c'])
        a_img, a_cap, o_cap = model.interaction_with_caption(image_embeds=v, caption_embeds=c,
                                                             key_padding_mask=~batch['mask_c'].bool(), video_mask=vm)
        feat_v = model.get_features(a_img.transpose(0, 1).contiguous(), None,
                                    vis_mask=batch['video_mask'].unsqueeze(-1))
    losses = model(batch['video'], batch['video_mask'],
                   [Enc(batch['ids_s'], batch['mask_s']), Enc(batch['ids_t'], batch['mask_t'])],
                   captions=Enc(batch['ids_c'], batch['mask_c']), idx=batch['idx'])
    sum(losses).backward()
    out = {k: v_.numpy() for k, v_ in batch.items()}
    out.update({'vision_embeds': v.numpy(), 'caption_hidden': c.numpy(), 'after_image': a_img.numpy(),
                'after_caption': a_cap.numpy(), 'ori_caption': o_cap.numpy(), 'feat_v': feat_v.numpy(),
                'losses': np.array([float(l) for l in losses], dtype=np.float64)})
    out.update(sd_np(model))
    out.update(grads_np(model))
    np.savez_compressed(os.path.join(OUT, 'video_tiny_w1.npz'), **out)
    print('video_tiny_w1 losses', out['losses'])


# ----------------------------------------------------------------------------- full-geometry fixtures
B16_VIT_1L = dict(embed_dim=512, image_resolution=384, vision_layers=1, vision_width=768, vision_patch_size=16,
                  context_length=8, vocab_size=32, transformer_width=64, transformer_heads=1, transformer_layers=1)
B16_VIT = dict(B16_VIT_1L, vision_layers=12)
MBERT = dict(vocab_size=119547, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
             max_position_embeddings=512, hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
FULL_CFG = {'embed_dim': 256, 'temp': 0.07, 'use_one_cl_proj_only': False, 'caption_encoder_name': 'mbert',
            'num_queries': 4, 'caption_ca_layer': 3, 'caption_interaction_layer': 2, 'weight_reg_loss': 0.01,
            'weight_caption_loss': 0.01, 'weight_dstl_loss': 0.5, 'weight_cv_loss': 0.01}      # = leccr_amd.synthetic.MULTI30K_CFG

# The geometry the reference's build_clip_encoder hard-codes (models/xvlm.py:83-88: clip.load('ViT-B/32'); clip/model.py:211-228
# gives (384 // 32)^2 + 1 = 145 tokens and 3 * 32 * 32 = 3072-wide patch rows) - what run_multi30k.sh really instantiates.
B32_VIT_1L = dict(B16_VIT_1L, vision_patch_size=32)
B32_VIT = dict(B32_VIT_1L, vision_layers=12)
FULL_GEOMETRY = {'b16': (B16_VIT_1L, B16_VIT, 51, 61, [0, 1, 288, 575, 576]),
                 'b32': (B32_VIT_1L, B32_VIT, 151, 161, [0, 1, 72, 143, 144])}


def fixture_vit_b16_1layer(clip_model, tag='b16'):
    """The reference's CLIP ViT at the REAL B/16 geometry (577 tokens, width 768, 12 heads, patch 16, res 384;
    clip/model.py:408-445), one layer, b = 2: checks the 577-token tiling of the production kernels.  Weights and inputs
    come from tests.helpers.seeded_fill_ / a seeded generator (rebuilt by the tests), outputs are stored as slices.
    tag 'b32': the same at the reference's hard-coded default geometry (145 tokens, patch 32)."""
    from tests.golden_slices import B16_GRAD_SLICES
    from tests.helpers import seeded_fill_
    geo, _, seed, _, rows = FULL_GEOMETRY[tag]
    torch.manual_seed(0)
    clip = clip_model.CLIP(**geo).float().eval()
    sums = seeded_fill_(clip.visual.named_parameters(), seed)
    g = torch.Generator().manual_seed(seed + 1)
    image = torch.randn(2, 3, 384, 384, generator=g)
    cls, hidden = clip.encode_image(image, return_hidden=True)
    trunk = clip.visual(image)
    probe = torch.randn(hidden.shape, generator=g)
    (hidden * probe).sum().backward()
    out = {'hidden_rows': hidden[:, rows, :].detach().numpy(), 'rows': np.array(rows), 'hidden_cols16': hidden[:, :, ::16].detach().numpy(),
           'trunk_cols16': trunk[:, :, ::16].detach().numpy(), 'hidden_sq_sum': np.array(float((hidden.double() ** 2).sum())),
           'wsum_names': np.array(sorted(sums)), 'wsum': np.array([sums[k] for k in sorted(sums)])}
    named = dict(clip.visual.named_parameters())
    for k, sl in B16_GRAD_SLICES.items():
        gk = named[k].grad
        out['g/' + k] = (gk if sl is None else gk[sl]).detach().numpy()
        out['gn/' + k] = np.array(float(gk.double().norm()))
    np.savez_compressed(os.path.join(OUT, f'vit_{tag}_1layer.npz'), **out)
    print(f'vit_{tag}_1layer', hidden.shape, float(hidden.abs().mean()), os.path.getsize(os.path.join(OUT, f'vit_{tag}_1layer.npz')))


def fixture_full_step_b16(clip_model, xvlm, tag='b16'):
    """tag 'b32': the reference's hard-coded default tower (145 tokens, patch 32) -> full_step_b32.npz.
    SURVEY 8(c) `full_step_b16`: the reference RetrievalModel at the REAL ViT-B/16@384 + mBERT geometry (12 + 12 layers,
    vocab 119 547, 4 queries, 3 + 2 + 2 interaction layers), b = 4, L_t = L_c = 12, seeded weights (tests.helpers.seeded_fill_):
    its five losses, the three feature matrices and norms / slices of a few gradients.  No weight payload: the GPU test
    rebuilds the same weights from the seed and must reproduce these numbers with the HIP model directly."""
    from tests.helpers import full_step_batch, seeded_fill_

    _, geo, _, seed, _ = FULL_GEOMETRY[tag]

    def _clip(config=None):
        return clip_model.CLIP(**geo).float(), geo['embed_dim']

    def _bert(config=None):
        from transformers import BertConfig, BertModel
        return BertModel(BertConfig(**MBERT)), MBERT['hidden_size']

    xvlm.build_clip_encoder = _clip
    xvlm.build_text_encoder = _bert
    from models.model_retrieval_caption import RetrievalModel
    torch.manual_seed(seed)
    model = RetrievalModel(dict(FULL_CFG))
    model.init_caption_encoder()
    sums = seeded_fill_(model.named_parameters(), seed)
    model.eval()
    batch = full_step_batch(seed0=seed + 1)
    with torch.no_grad():
        v, _ = model.get_vision_embeds(batch['image'])
        c = model.get_caption_embeds(batch['ids_c'], batch['mask_c'])
        a_img, a_cap, o_cap = model.interaction_with_caption(image_embeds=v, caption_embeds=c,
                                                             key_padding_mask=~batch['mask_c'].bool())
        feat_v = model.get_features(a_img.transpose(0, 1).contiguous(), None)
        feat_s = model.get_features(None, model.get_text_embeds(batch['ids_s'], batch['mask_s']))
        feat_t = model.get_features(None, model.get_text_embeds(batch['ids_t'], batch['mask_t']))
    losses = run_forward(model, batch)
    sum(losses).backward()
    named = dict(model.named_parameters())
    out = {'losses': np.array([float(l) for l in losses], dtype=np.float64), 'feat_v': feat_v.numpy(), 'feat_s': feat_s.numpy(),
           'feat_t': feat_t.numpy(), 'vision_cols32': v[:, :, ::32].numpy(), 'caption_hidden_cols32': c[:, :, ::32].numpy(),
           'after_caption': a_cap.numpy(), 'ori_caption_cols8': o_cap[:, :, ::8].numpy(),
           'wsum_names': np.array(sorted(sums)), 'wsum': np.array([sums[k] for k in sorted(sums)])}
    from tests.golden_slices import FULL_STEP_GRAD_SLICES as keep
    for k, sl in keep.items():
        gk = named[k].grad
        out['g/' + k] = np.asarray((gk if sl is None else gk[sl]).detach().numpy())
        out['gn/' + k] = np.array(float(gk.double().norm()))
    np.savez_compressed(os.path.join(OUT, f'full_step_{tag}.npz'), **out)
    print(f'full_step_{tag} losses', out['losses'], os.path.getsize(os.path.join(OUT, f'full_step_{tag}.npz')))


def fixture_pos_embed(clip_model):
    """interpolate_pos_embed (clip/model.py:456-481) on seeded tables: the ViT-B/16 (14x14 -> 24x24), B/32 (7x7 -> 12x12)
    and L/14 (16x16 -> 27x27) cases of build_model at its hard-coded 384 resolution."""
    out = {}
    g = torch.Generator().manual_seed(71)
    for name, g0, g1, d in (('b16', 14, 24, 48), ('b32', 7, 12, 48), ('l14', 16, 27, 32)):
        pe = torch.randn(1, 1 + g0 * g0, d, generator=g)
        out[name + '/in'] = pe.numpy()
        out[name + '/out'] = clip_model.interpolate_pos_embed(pe, num_patches=g1 * g1, num_extra_tokens=1).numpy()
    np.savez_compressed(os.path.join(OUT, 'pos_embed.npz'), **out)
    print('pos_embed', {k: v.shape for k, v in out.items()})


def caption_store_spec():
    """A small annotation store in the reference's layout (Multi30K-style and mscoco-style), as {relative path: text}."""
    ids = ['1000092795', '10002456', 'a-b_c7', '33']
    files = {'img_id/train_id.txt': '\n'.join(ids) + '\n', 'img_id/val_id.txt': '\n'.join(ids[1:]) + '\n',
             'img_id/test_id_2016.txt': '\n'.join(ids[:3]) + '\n'}
    caps_en = ['Two young, White males are outside near many bushes.', 'A man in an orange hat starring at something!',
               "A <person> is sitting; it's a dog/cat - really?  (yes)", 'x ' * 40 + 'end', 'Several men in hard hats are operating a giant pulley system.']
    caps_de = ['Zwei junge weiße Männer sind im Freien in der Nähe vieler Büsche.', 'Ein Mann mit einem orangefarbenen Hut, der etwas anstarrt.',
               'Eine Person sitzt; es ist ein Hund/Katze - wirklich?', 'y ' * 35 + 'ende', 'Mehrere Männer mit Schutzhelmen bedienen ein Antriebsradsystem.']
    order = [(0, 0), (1, 1), (2, 2), (3, 3), (0, 4)]          # (image, caption): image 0 has two captions
    files['train_enc.txt'] = ''.join(f'{ids[i]}.jpg#enc#{k} {caps_en[c]}\n' for k, (i, c) in enumerate(order))
    files['train_enc2de.txt'] = ''.join(f'{ids[i]}.jpg#enc2de#{k} {caps_de[c]}\n' for k, (i, c) in enumerate(order))
    files['val_de.txt'] = ''.join(f'{ids[i]}.jpg#enc2de#{k} {caps_de[c]}\n' for k, (i, c) in enumerate([(1, 1), (2, 2), (1, 0), (3, 3)]))
    files['val_de_trans.txt'] = ''.join(f'{ids[i]}.jpg#enc#{k} {caps_en[c]}\n' for k, (i, c) in enumerate([(1, 1), (2, 2), (1, 0), (3, 3)]))
    for i in ids:
        files[f'gen/{i}.txt'] = f'a generated caption for image {i}.\n'
    # mscoco flavour: ids map to file names through img_id/image_ids.txt; generated captions are named after the FILE
    coco = {'c1': 'COCO_val2014_000000391895.jpg', 'c2': 'jpeg_COCO_000000522418.jpg'}
    files['coco/img_id/image_ids.txt'] = ''.join(f'{k} {v}\n' for k, v in coco.items())
    files['coco/img_id/train_id.txt'] = 'c1\nc2\n'
    files['coco/img_id/ja_val_id.txt'] = 'c2\nc1\n'
    files['coco/train_enc.txt'] = 'c1#enc#0 A woman riding a bike.\nc2#enc#1 A plate of food, with a fork.\n'
    files['coco/train_enc2ja.txt'] = 'c1#enc2ja#0 jitensha ni noru josei\nc2#enc2ja#1 fooku to ryouri no sara\n'
    files['coco/val_ja.txt'] = 'c2#enc2ja#0 ryouri no sara\nc1#enc2ja#1 jitensha\nc2#enc2ja#2 fooku\n'
    for k, v in coco.items():
        files['coco/gen/' + v.strip('.jpg') + '.txt'] = f'generated for {k}\n'
    return files, ids, coco


def fixture_caption_store():
    """The reference's re_train_dataset_caption / re_eval_dataset_caption (dataset/retrieval_dataset.py:30-264) and pre_caption
    (dataset/utils.py:31-55) run on a generated annotation store: what leccr_amd.captions must reproduce.  The class sources are
    compiled from the reference file with a namespace that supplies their imports (the `dataset` package itself needs
    torchvision); images are 4x4 JPEGs written here, `transform` returns the image size."""
    import ast
    import tempfile
    from PIL import Image
    from torch.utils.data import Dataset
    ns = {'re': __import__('re')}
    tree = ast.parse(open(os.path.join(REF, 'dataset/utils.py')).read())
    exec(compile(ast.Module(body=[n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == 'pre_caption'], type_ignores=[]),
                 'utils.py', 'exec'), ns)
    ns.update(os=os, np=np, torch=torch, time=__import__('time'), Image=Image, Dataset=Dataset, json=json)
    tree = ast.parse(open(os.path.join(REF, 'dataset/retrieval_dataset.py')).read())
    keep = [n for n in tree.body if getattr(n, 'name', None) in ('getVideoId', 're_train_dataset_caption', 're_eval_dataset_caption')]
    exec(compile(ast.Module(body=keep, type_ignores=[]), 'retrieval_dataset.py', 'exec'), ns)
    files, ids, coco = caption_store_spec()
    out = {'files': files, 'pre_caption': {}}
    for text in ["A <person>, sitting - on/a BENCH!!  (twice)", 'x ' * 40, 'Hello;World:~ok', "don't #stop*"]:
        out['pre_caption'][text] = ns['pre_caption'](text, 30)
    with tempfile.TemporaryDirectory() as d:
        for rel, text in files.items():
            os.makedirs(os.path.dirname(os.path.join(d, rel)), exist_ok=True)
            open(os.path.join(d, rel), 'w').write(text)
        os.makedirs(os.path.join(d, 'images'))
        for name in [i + '.jpg' for i in ids] + list(coco.values()):
            Image.new('RGB', (4, 4), (10, 20, 30)).save(os.path.join(d, 'images', name))
        tf = lambda im: im.size
        for tag, cfg, trains, ev in (
                ('m30k', dict(root_dir=d, image_root=os.path.join(d, 'images'), dataset='multi30k', generated_caption_dir=os.path.join(d, 'gen'),
                              generated_caption_type='text', train_file=['train_enc.txt', 'train_enc2de.txt']),
                 None, ('val_de.txt', os.path.join(d, 'val_de_trans.txt'))),
                ('coco', dict(root_dir=os.path.join(d, 'coco'), image_root=os.path.join(d, 'images'), dataset='mscoco',
                              generated_caption_dir=os.path.join(d, 'coco/gen'), generated_caption_type='text',
                              train_file=['train_enc.txt', 'train_enc2ja.txt']), None, ('val_ja.txt', None))):
            tr = ns['re_train_dataset_caption'](cfg, tf, 30)
            recs = []
            for i in range(len(tr)):
                size, caps, gen, img_idx, cap_id = tr[i]
                recs.append([list(size), caps, gen, img_idx, cap_id])
            e = ns['re_eval_dataset_caption'](cfg, ev[0], tf, 30, test_trans_file=ev[1], type='eval')
            erecs = [[list(e[i][0]), e[i][1], e[i][2]] for i in range(len(e))]
            out[tag] = {'train_len': len(tr), 'train_records': recs, 'languages': tr.languages, 'img_ids': tr.img_ids,
                        'eval_text': e.text, 'eval_image': e.image, 'txt2img': {str(k): v for k, v in e.txt2img.items()},
                        'img2txt': {str(k): v for k, v in e.img2txt.items()}, 'eval_cap_ids': e.cap_ids, 'eval_records': erecs,
                        'text_trans': getattr(e, 'text_trans', None)}
    json.dump(out, open(os.path.join(OUT, 'caption_store.json'), 'w'), indent=1, ensure_ascii=False)
    print('caption_store', {k: (v['train_len'], len(v['eval_text'])) for k, v in out.items() if k in ('m30k', 'coco')})


def main():
    if os.path.dirname(os.path.dirname(OUT)) not in sys.path:
        sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))          # tests.helpers (seeded_fill_)
    only = set(sys.argv[1:])
    if only == {'w4'}:
        fixture_model_w4()
        return
    torch.set_num_threads(4)
    clip_model, xvlm = import_reference()
    if not dist.is_initialized():
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29610')
        dist.init_process_group('gloo', rank=0, world_size=1)
    if only:            # python tests/golden/make_golden.py full  -> only the full-geometry fixtures (minutes of CPU)
        assert only <= {'full', 'captions', 'b32', 'clipcap', 'w4'}, only
        if only == {'clipcap'}:
            fixture_model_clipcap(clip_model, xvlm)
            dist.destroy_process_group()
            return
        if only == {'captions'}:
            fixture_caption_store()
            dist.destroy_process_group()
            return
        if only == {'b32'}:
            torch.set_num_threads(8)
            fixture_vit_b16_1layer(clip_model, 'b32')
            fixture_full_step_b16(clip_model, xvlm, 'b32')
            dist.destroy_process_group()
            return
        torch.set_num_threads(8)
        fixture_pos_embed(clip_model)
        fixture_vit_b16_1layer(clip_model)
        fixture_full_step_b16(clip_model, xvlm)
        fixture_vit_b16_1layer(clip_model, 'b32')
        fixture_full_step_b16(clip_model, xvlm, 'b32')
        dist.destroy_process_group()
        return
    fixture_vit(clip_model)
    fixture_bert()
    fixture_model_w1(clip_model, xvlm)
    fixture_model_clipcap(clip_model, xvlm)
    fixture_losses(clip_model, xvlm)
    fixture_itm_eval()
    fixture_video_w1()
    fixture_pos_embed(clip_model)
    fixture_vit_b16_1layer(clip_model)
    fixture_full_step_b16(clip_model, xvlm)
    fixture_vit_b16_1layer(clip_model, 'b32')
    fixture_full_step_b16(clip_model, xvlm, 'b32')
    fixture_caption_store()
    dist.destroy_process_group()
    fixture_model_w2()
    fixture_model_w4()


if __name__ == '__main__':
    main()
