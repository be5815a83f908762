"""Pin oracle/ (the CPU restatement) to fixtures captured from the reference (tests/golden/make_golden.py)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import leccr_oracle as O
from tests.helpers import GOLDEN, T, grad_close, load_npz, max_err, rel_err, weights

TINY_CFG = {'weight_reg_loss': 0.01, 'weight_caption_loss': 0.01, 'weight_dstl_loss': 0.5, 'weight_cv_loss': 0.01}
FWD_TOL = 2e-5      # fp32, identical math up to summation order
GRAD_TOL = 2e-4     # relative to the tensor's max |grad|


def test_vit_trunk_and_hidden():
    z = load_npz('vit_tiny.npz')
    sd = weights(z, requires_grad=True)
    image = T(z['image']).requires_grad_(True)
    trunk = O.vit_trunk(sd, '', image)
    assert max_err(trunk, z['trunk']) < FWD_TOL
    hidden = O.encode_image(sd, '', image)
    assert max_err(hidden, z['hidden']) < FWD_TOL
    assert max_err(hidden[:, 0], z['cls']) < FWD_TOL
    (hidden * T(z['probe'])).sum().backward()
    assert rel_err(image.grad, z['g_image']) < GRAD_TOL
    checked = 0
    for k, v in z.items():
        if k.startswith('g/'):
            assert grad_close(sd[k[2:]].grad, v, GRAD_TOL), k
            checked += 1
    assert checked > 15


def test_bert():
    z = load_npz('bert_tiny.npz')
    sd = weights(z, requires_grad=True)
    h = O.bert_forward(sd, '', T(z['ids']), T(z['mask']), heads=2)
    assert max_err(h, z['last_hidden_state']) < FWD_TOL
    (h * T(z['probe'])).sum().backward()
    checked = 0
    for k, v in z.items():
        if k.startswith('g/'):
            assert grad_close(sd[k[2:]].grad, v, GRAD_TOL), k
            checked += 1
    assert checked > 30


def _batch(z):
    return dict(image=T(z['image']), ids_s=T(z['ids_s']), mask_s=T(z['mask_s']), ids_t=T(z['ids_t']),
                mask_t=T(z['mask_t']), ids_c=T(z['ids_c']), mask_c=T(z['mask_c']), idx=T(z['idx']))


def test_interaction_and_features():
    z = load_npz('model_tiny_w1.npz')
    sd = weights(z)
    b = _batch(z)
    vis = O.encode_image(sd, 'clip_encoder.visual.', b['image'])
    assert max_err(vis, z['vision_embeds']) < FWD_TOL
    cap = O.bert_forward(sd, 'text_encoder.', b['ids_c'], b['mask_c'], heads=2)
    assert max_err(cap, z['caption_hidden']) < FWD_TOL
    a_img, a_cap, o_cap = O.interaction_with_caption(sd, vis, cap, ~b['mask_c'].bool())
    # reference returns LND
    assert max_err(a_img.transpose(0, 1), z['after_image']) < FWD_TOL
    assert max_err(a_cap.transpose(0, 1), z['after_caption']) < FWD_TOL
    assert max_err(o_cap.transpose(0, 1), z['ori_caption']) < FWD_TOL
    assert max_err(O.get_features(sd, image_embeds=a_img), z['feat_v']) < FWD_TOL
    ts = O.bert_forward(sd, 'text_encoder.', b['ids_s'], b['mask_s'], heads=2)
    assert max_err(O.get_features(sd, text_embeds=ts), z['feat_s']) < FWD_TOL


def test_full_forward_backward_w1():
    z = load_npz('model_tiny_w1.npz')
    sd = weights(z, requires_grad=True)
    b = _batch(z)
    losses = O.retrieval_forward(sd, TINY_CFG, b['image'], b['ids_s'], b['mask_s'], b['ids_t'], b['mask_t'],
                                 b['ids_c'], b['mask_c'], b['idx'], text_heads=2)
    got = np.array([float(l) for l in losses])
    assert np.max(np.abs(got - z['losses'])) < 1e-5, (got, z['losses'])
    sum(losses).backward()
    checked = 0
    for k, v in z.items():
        if k.startswith('g/'):
            name = k[2:]
            assert sd[name].grad is not None, name
            assert grad_close(sd[name].grad, v, 5e-4), name
            checked += 1
    assert checked > 100
    # parameters the reference leaves without gradient stay without gradient here too
    for name in ('itm_head.0.weight', 'crossattn.layers.0.norm1.weight', 'text_encoder.pooler.dense.weight'):
        assert sd[name].grad is None, name


@pytest.mark.parametrize('name', ['itc_idx', 'itc_noidx', 'cap_itc', 'dstl', 'cv', 'reg'])
def test_losses(name):
    z = load_npz('losses.npz')
    sd = weights(z, requires_grad=True)
    t = {k: T(z[k]).requires_grad_(True) for k in ('fv', 'fs', 'ft', 'cq', 'cap', 'img')}
    idx = T(z['idx'])
    cq_b, cap_b = t['cq'].transpose(0, 1), t['cap'].transpose(0, 1)      # fixtures are LND
    loss = {'itc_idx': lambda: O.contrastive_loss(t['fv'], t['fs'], sd['temp'], idx),
            'itc_noidx': lambda: O.contrastive_loss(t['fv'], t['ft'], sd['temp'], None),
            'cap_itc': lambda: O.caption_contrastive_loss(cq_b, t['fs'], sd['temp']),
            'dstl': lambda: O.dstl_loss(t['fv'], cq_b, t['fs'], t['ft']),
            'cv': lambda: O.caption_vision_loss(sd, cap_b, t['img'], idx),
            'reg': lambda: O.caption_regularization(cap_b)}[name]()
    assert abs(float(loss) - float(z[f'{name}/loss'])) < 1e-5 * max(1.0, abs(float(z[f'{name}/loss'])))
    loss.backward()
    for k, v in z.items():
        if k.startswith(f'{name}/g_'):
            key = k[len(name) + 3:]
            src = t[key] if key in t else sd[key]
            g = src.grad if src.grad is not None else torch.zeros_like(src)
            assert max_err(g, v) < 1e-5 + GRAD_TOL * float(np.max(np.abs(v))), k


def test_itm_eval_bit_exact():
    cases = json.load(open(os.path.join(GOLDEN, 'itm_eval.json')))
    for name, c in cases.items():
        s = np.load(os.path.join(GOLDEN, f'itm_scores_{name}.npy'))
        ni, nt = s.shape
        per = c['per']
        img2txt = {i: list(range(i * per, (i + 1) * per)) for i in range(ni)}
        txt2img = {t: t // per for t in range(nt)}
        res = O.itm_eval(s, s.T.copy(), txt2img, img2txt)
        assert set(res) == set(c['result'])
        for k, v in c['result'].items():
            assert float(res[k]) == v, (name, k, res[k], v)     # integer ranks -> exact percentages


# ----------------------------------------------------------------------------------------- video model
VIDEO_CFG = dict(TINY_CFG, vision_width=64, vision_layer=2)


def _vbatch(z):
    return dict(video=T(z['video']), video_mask=T(z['video_mask']), ids_s=T(z['ids_s']), mask_s=T(z['mask_s']),
                ids_t=T(z['ids_t']), mask_t=T(z['mask_t']), ids_c=T(z['ids_c']), mask_c=T(z['mask_c']), idx=T(z['idx']))


def test_video_interaction_and_features():
    """Oracle vs the reference's video RetrievalModel (fixture video_tiny_w1.npz: ragged clips of 3..8 frames)."""
    z = load_npz('video_tiny_w1.npz')
    sd = weights(z)
    b = _vbatch(z)
    assert 3 <= int(b['video_mask'].sum(1).min()) < int(b['video_mask'].sum(1).max()) == 8
    vid, vkpm = O.video_encode(sd, b['video'], b['video_mask'])
    # padded frames are keys nobody attends to, but their own rows are still computed: compare every row
    assert max_err(vid, z['vision_embeds']) < FWD_TOL
    cap = O.bert_forward(sd, 'text_encoder.', b['ids_c'], b['mask_c'], heads=2)
    a_vid, a_cap, o_cap = O.video_interaction_with_caption(sd, vid, cap, ~b['mask_c'].bool(), vkpm)
    assert max_err(a_vid.transpose(0, 1), z['after_image']) < FWD_TOL
    assert max_err(a_cap.transpose(0, 1), z['after_caption']) < FWD_TOL
    assert max_err(o_cap.transpose(0, 1), z['ori_caption']) < FWD_TOL
    assert max_err(O.video_get_features(sd, a_vid, b['video_mask']), z['feat_v']) < FWD_TOL


def test_video_full_forward_backward_w1():
    z = load_npz('video_tiny_w1.npz')
    sd = weights(z, requires_grad=True)
    b = _vbatch(z)
    losses = O.video_retrieval_forward(sd, VIDEO_CFG, b['video'], b['video_mask'], b['ids_s'], b['mask_s'], b['ids_t'],
                                       b['mask_t'], b['ids_c'], b['mask_c'], b['idx'], text_heads=2)
    got = np.array([float(l) for l in losses])
    assert np.max(np.abs(got - z['losses'])) < 1e-5, (got, z['losses'])
    sum(losses).backward()
    checked = 0
    for k, v in z.items():
        if k.startswith('g/'):
            name = k[2:]
            assert sd[name].grad is not None, name
            assert grad_close(sd[name].grad, v, 5e-4), name
            checked += 1
    assert checked > 90
    assert any(k.startswith('g/vid_encoder.layers.1.') for k in z) and not any(k.startswith('w/clip_encoder.') for k in z)
