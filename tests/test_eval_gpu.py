"""itm_eval on the GPU: bit-exact with the reference's numpy itm_eval on the committed tie-free fixtures, and with the
oracle on a larger 1000 x 5000 (MSCOCO-1k shaped) score matrix; evaluation_coarse end to end on a synthetic loader."""
import json
import os
import types

import numpy as np
import pytest
import torch

from tests.helpers import GOLDEN

pytestmark = pytest.mark.gpu


def test_itm_eval_bit_exact_with_reference_fixture():
    from leccr_amd.evaluation import itm_eval
    cases = json.load(open(os.path.join(GOLDEN, 'itm_eval.json')))
    for name, c in cases.items():
        s = np.load(os.path.join(GOLDEN, f'itm_scores_{name}.npy'))
        ni, nt = s.shape
        per = c['per']
        img2txt = {i: list(range(i * per, (i + 1) * per)) for i in range(ni)}
        txt2img = {t: t // per for t in range(nt)}
        res = itm_eval(s, s.T.copy(), txt2img, img2txt)
        assert set(res) == set(c['result'])
        for k, v in c['result'].items():
            assert float(res[k]) == v, (name, k, res[k], v)


def test_itm_eval_full_size_matches_oracle():
    from leccr_amd.evaluation import itm_eval
    from oracle import leccr_oracle as O
    rng = np.random.default_rng(3)
    ni, per = 1000, 5
    nt = ni * per
    s = rng.permutation(ni * nt).reshape(ni, nt).astype(np.float32) / (ni * nt)       # tie-free
    for i in range(ni):
        s[i, i * per + rng.integers(per)] += rng.random() * 0.9
    img2txt = {i: list(range(i * per, (i + 1) * per)) for i in range(ni)}
    txt2img = {t: t // per for t in range(nt)}
    got = itm_eval(torch.from_numpy(s).cuda(), torch.from_numpy(s.T.copy()).cuda(), txt2img, img2txt)
    want = O.itm_eval(s, s.T.copy(), txt2img, img2txt)
    assert {k: float(v) for k, v in got.items()} == {k: float(v) for k, v in want.items()}


def test_evaluation_coarse_synthetic_loader():
    """evaluation_coarse -> scores equal the oracle's features' dot products (f32 path, tiny geometry)."""
    from leccr_amd.evaluation import evaluation_coarse
    from oracle import leccr_oracle as O
    from tests.helpers import load_npz, weights
    from tests.test_model_gpu import MODEL_VIT, TINY_BERT, build
    z = load_npz('model_tiny_w1.npz')
    sd = weights(z)
    model = build(dict(clip_geometry=MODEL_VIT, text_config=TINY_BERT, compute_dtype='fp32'), sd)
    ids_s, m_s = torch.from_numpy(z['ids_s']), torch.from_numpy(z['mask_s'])
    ids_c, m_c = torch.from_numpy(z['ids_c']), torch.from_numpy(z['mask_c'])
    image = torch.from_numpy(z['image'])

    class Tok:                                   # the tokenizer interface evaluation_coarse uses
        def __call__(self, text, **kw):
            ids, m = (ids_s, m_s) if text[0][0] == 't' else (ids_c, m_c)
            sel = torch.tensor([int(t[1:]) for t in text])
            o = types.SimpleNamespace(input_ids=ids[sel], attention_mask=m[sel])
            o.to = lambda dev: types.SimpleNamespace(input_ids=o.input_ids.to(dev), attention_mask=o.attention_mask.to(dev))
            return o

    class Loader(list):
        dataset = types.SimpleNamespace(text=[f't{i}' for i in range(8)])
    loader = Loader([(image[i:i + 4], [f'c{j}' for j in range(i, i + 4)], None) for i in (0, 4)])
    i2t, t2i = evaluation_coarse(model, loader, Tok(), 'cuda', {'batch_size_test_text': 3, 'max_tokens': 200})
    vis = O.encode_image(sd, 'clip_encoder.visual.', image)
    cap = O.bert_forward(sd, 'text_encoder.', ids_c, m_c, 2)
    a_img, _, _ = O.interaction_with_caption(sd, vis, cap, ~m_c.bool())
    fv = O.get_features(sd, image_embeds=a_img)
    ft = O.get_features(sd, text_embeds=O.bert_forward(sd, 'text_encoder.', ids_s, m_s, 2))
    ref = (fv @ ft.t()).numpy()
    assert i2t.shape == (8, 8) and np.max(np.abs(i2t - ref)) < 1e-4 and np.array_equal(t2i, i2t.T)
    assert np.array_equal(np.argsort(-i2t, 1), np.argsort(-ref, 1))             # identical rankings


def test_evaluation_coarse_video_double_sim():
    """evaluation_coarse_video (video_Retrieval_caption_double_sim.py:94-190) == the same formula on the oracle's features."""
    from leccr_amd.evaluation import evaluation_coarse_video
    from oracle import leccr_oracle as O
    from tests.helpers import load_npz, weights
    from tests.test_model_gpu import TINY_BERT, build_video
    z = load_npz('video_tiny_w1.npz')
    sd = weights(z)
    model = build_video(dict(text_config=TINY_BERT, compute_dtype='fp32'), sd)
    ids_s, m_s = torch.from_numpy(z['ids_s']), torch.from_numpy(z['mask_s'])
    ids_c, m_c = torch.from_numpy(z['ids_c']), torch.from_numpy(z['mask_c'])
    video, vmask = torch.from_numpy(z['video']), torch.from_numpy(z['video_mask'])

    class Tok:
        def __call__(self, text, **kw):
            ids, m = (ids_s, m_s) if text[0][0] == 't' else (ids_c, m_c)
            sel = torch.tensor([int(t[1:]) for t in text])
            o = types.SimpleNamespace(input_ids=ids[sel], attention_mask=m[sel])
            o.to = lambda dev: types.SimpleNamespace(input_ids=o.input_ids.to(dev), attention_mask=o.attention_mask.to(dev))
            return o

    class Loader(list):
        dataset = types.SimpleNamespace(text=[f't{i}' for i in range(8)])
    loader = Loader([(video[a:b], vmask[a:b], [f'c{j}' for j in range(a, b)], None) for a, b in ((0, 3), (3, 8))])
    i2t, t2i = evaluation_coarse_video(model, loader, Tok(), 'cuda', {'batch_size_test_text': 3, 'max_tokens': 200})
    vid, vkpm = O.video_encode(sd, video, vmask)
    cap = O.bert_forward(sd, 'text_encoder.', ids_c, m_c, 2)
    a_vid, a_cap, _ = O.video_interaction_with_caption(sd, vid, cap, ~m_c.bool(), vkpm)
    fv = O.video_get_features(sd, a_vid, vmask)
    ft = O.get_features(sd, text_embeds=O.bert_forward(sd, 'text_encoder.', ids_s, m_s, 2))
    cq = torch.nn.functional.linear(a_cap, sd['caption_proj1.weight'], sd['caption_proj1.bias'])       # [b, n, D]
    norm = lambda s: (s - s.min()) / (s - s.min()).max()
    ref = 0.9 * norm(fv @ ft.t()) + 0.1 * norm(torch.einsum('ind,jd->inj', cq, ft).max(1)[0])
    assert i2t.shape == (8, 8) and np.max(np.abs(i2t - ref.numpy())) < 1e-4 and np.array_equal(t2i, i2t.T)
    assert np.array_equal(np.argsort(-i2t, 1), np.argsort(-ref.numpy(), 1))
