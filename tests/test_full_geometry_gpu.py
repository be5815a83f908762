"""Parity at the BENCHMARKED geometry (BASELINE config B: ViT-B/16 @384, 577 tokens, 12 + 12 layers, mBERT vocabulary) on MI355X.

1. `vit_b16_1layer` / `full_step_b16`: the HIP model with the seeded weights reproduces numbers captured from the REFERENCE's own
   classes at the real geometry (tests/golden/make_golden.py) - fp32 path to the north-star 1e-3, bf16 path
   against the same numbers.
2. The full model on a Multi30K-shaped batch (L_t 32, L_c 64) against the CPU oracle, forward and every parameter gradient.
3. The bf16 path's gradients, tensor by tensor, against the oracle's bf16-storage noise model (oracle.lowp) and, at temp = 1
   (no 1/0.07 amplification of feature rounding), against a uniform bound - replaces the old "rel <= 0.7" allowance.
"""
import os

import numpy as np
import pytest
import torch

from tests.helpers import (B32_GEOMETRY, T, check_weight_sums, noise_above_ceiling, write_noise_table, full_geometry_model, full_step_batch, grad_close, grad_rows, load_npz, max_err,
                           outside_noise_model, seeded_fill_)

pytestmark = pytest.mark.gpu
DEV = 'cuda'
WL = {'weight_reg_loss': 0.01, 'weight_caption_loss': 0.01, 'weight_dstl_loss': 0.5, 'weight_cv_loss': 0.01}


class Enc:
    def __init__(self, ids, mask):
        self.input_ids, self.attention_mask = ids, mask


def _forward(model, b):
    g = lambda k: b[k].to(DEV)
    return model(g('image'), [Enc(g('ids_s'), g('mask_s')), Enc(g('ids_t'), g('mask_t'))],
                 captions=Enc(g('ids_c'), g('mask_c')), idx=g('idx'))


# ----------------------------------------------------------------------------------------- 1. reference-captured, real geometry
GEOMETRIES = {'b16': dict(geometry=None, vit_seed=51, seed=61, seed0=62, tokens=577),
              # what run_multi30k.sh really instantiates: models/xvlm.py:83-88 hard-codes clip.load('ViT-B/32') -> 145 tokens
              # (= 2*64 + 17), 3072-wide patch rows (clip/model.py:211-228); also this package's default clip_name
              'b32': dict(geometry=B32_GEOMETRY, vit_seed=151, seed=161, seed0=162, tokens=145)}


@pytest.mark.parametrize('tag', ['b16', 'b32'])
@pytest.mark.parametrize('dtype', ['fp32', 'bf16'])
def test_vit_b16_one_layer_real_geometry_vs_reference(dtype, tag):
    """577 tokens x 768 wide x 12 heads of 64, patch 16 at 384, one layer, b = 2: the production ViT kernels (patchify, MFMA /
    generic GEMMs, flash attention with the 577 = 9*64+1 tail, LayerNorm) against the reference's CLIP class.  'b32': the
    reference's hard-coded default tower - 145 tokens (two 64-key tiles + a 17-key tail), patch 32 (K = 3072 patch rows)."""
    from tests.golden_slices import B16_GRAD_SLICES
    G = GEOMETRIES[tag]
    z = load_npz(f'vit_{tag}_1layer.npz')
    model, _ = full_geometry_model(dtype, vision_layers=1, text_config=MID_BERT, geometry=G['geometry'])       # (the text tower is not used here)
    vis = model.clip_encoder.visual
    check_weight_sums(seeded_fill_(vis.named_parameters(), G['vit_seed']), z)
    model = model.to(DEV).eval()
    g = torch.Generator().manual_seed(G['vit_seed'] + 1)
    image = torch.randn(2, 3, 384, 384, generator=g)
    hidden, _ = model.get_vision_embeds(image.to(DEV))
    assert hidden.shape == (2, G['tokens'], 512)
    tol = 1e-3 if dtype == 'fp32' else 6e-2            # bf16: |hidden| ~ 0.7 +- 8-bit mantissas through 8 roundings
    rows = [int(r) for r in z['rows']]
    assert max_err(hidden[:, rows, :].float(), z['hidden_rows']) < tol
    assert max_err(hidden[:, :, ::16].float(), z['hidden_cols16']) < tol
    sq = float((hidden.double() ** 2).sum())
    assert abs(sq - float(z['hidden_sq_sum'])) < (1e-4 if dtype == 'fp32' else 5e-3) * float(z['hidden_sq_sum'])
    probe = torch.randn(hidden.shape, generator=g)
    model.arena.zero_grad()
    (hidden.float() * probe.to(DEV)).sum().backward()
    named = dict(vis.named_parameters())
    for k, sl in B16_GRAD_SLICES.items():
        gk = named[k].grad
        gn = float(z['gn/' + k])
        assert abs(float(gk.double().norm()) - gn) <= (2e-3 if dtype == 'fp32' else 3e-2) * gn, (k, float(gk.double().norm()), gn)
        got = gk if sl is None else gk[sl]
        if dtype == 'fp32':
            assert grad_close(got, z['g/' + k], 2e-3), k
        else:       # slice-level agreement: relative L2 error of the kept slice
            want = T(z['g/' + k]).double()
            rel = float((got.detach().cpu().double() - want).norm() / want.norm())
            assert rel <= 5e-2, (k, rel)


@pytest.mark.parametrize('tag', ['b16', 'b32'])
@pytest.mark.parametrize('dtype', ['fp32', 'bf16'])
def test_full_step_b16_mbert_vs_reference(dtype, tag):
    """SURVEY 8(c) `full_step_b16`: the full 12 + 12-layer model, b = 4, against the five losses, three feature matrices and
    gradient slices the REFERENCE produced with the same seeded weights.  'b32' = `full_step_b32`: the same with the tower the
    reference's build_clip_encoder hard-codes (ViT-B/32 @384: what "drops into run_multi30k.sh unchanged" runs)."""
    from tests.golden_slices import FULL_STEP_GRAD_SLICES
    G = GEOMETRIES[tag]
    z = load_npz(f'full_step_{tag}.npz')
    model, sums = full_geometry_model(dtype, seed=G['seed'], geometry=G['geometry'])
    check_weight_sums(sums, z)
    model = model.to(DEV).eval()
    b = full_step_batch(seed0=G['seed0'])
    with torch.no_grad():
        v, _ = model.get_vision_embeds(b['image'].to(DEV))
        c = model.get_caption_embeds(b['ids_c'].to(DEV), b['mask_c'].to(DEV))
        a_img, a_cap, o_cap = model.interaction_with_caption(image_embeds=v, caption_embeds=c,
                                                             key_padding_mask=~b['mask_c'].to(DEV).bool())
        feat_v = model.get_features(a_img.transpose(0, 1).contiguous(), None)
        feat_s = model.get_features(None, model.get_text_embeds(b['ids_s'].to(DEV), b['mask_s'].to(DEV)))
    ftol, atol = (1e-4, 1e-3) if dtype == 'fp32' else (2e-2, 1e-1)
    assert max_err(v[:, :, ::32].float(), z['vision_cols32']) < atol
    assert max_err(c[:, :, ::32].float(), z['caption_hidden_cols32']) < atol
    assert max_err(a_cap.float(), z['after_caption']) < atol
    assert max_err(feat_v, z['feat_v']) < ftol and max_err(feat_s, z['feat_s']) < ftol
    losses = _forward(model, b)
    got = np.array([float(l) for l in losses])
    # north-star bound: logits within 1e-3 in fp32.  bf16: feature rounding (2^-9) x 1/temp (14) on logits of G = 4 rows
    ltol = 1e-3 if dtype == 'fp32' else 3e-2
    assert np.max(np.abs(got - z['losses'])) < ltol, (got, z['losses'])
    model.arena.zero_grad()
    sum(losses).backward()
    named = dict(model.named_parameters())
    noise = None
    if dtype == 'bf16':     # what bf16 storage does to each of these gradients in the oracle (same weights, same batch)
        torch.set_num_threads(16)
        bb = dict(b)
        _, noise = oracle_step(model.state_dict(), bb, 12, lowp=True)
    bad = []
    for k, sl in FULL_STEP_GRAD_SLICES.items():
        gk = named[k].grad
        gn = float(z['gn/' + k])
        got_n = float(gk.double().norm())
        cut = lambda t: (t if sl is None else t[sl]).detach().cpu().double().reshape(-1)
        want = T(z['g/' + k]).double().reshape(-1)
        err = float((cut(gk) - want).norm())
        if dtype == 'fp32':         # the reference's own gradient values
            ok = abs(got_n - gn) <= 2e-3 * gn and err <= 2e-3 * float(want.norm())
        else:                       # within 3x the bf16-storage noise of THIS slice (+ 2e-3 of it)
            nz = float((cut(noise[k].grad) - want).norm())
            ok = err <= 3.0 * nz + 2e-3 * float(want.norm()) and abs(got_n - gn) <= 3.0 * abs(float(noise[k].grad.double().norm()) - gn) + 0.02 * gn
        if not ok:
            bad.append((k, got_n, gn, err / float(want.norm())))
    assert not bad, bad


# ----------------------------------------------------------------------------------------- 2. full model vs oracle, Multi30K shapes
grad_table = grad_rows


def oracle_step(sd_src, batch, text_heads, lowp=False, temp=None):
    from oracle import leccr_oracle as O
    sd = {k: v.detach().cpu().float().clone().requires_grad_(v.is_floating_point()) for k, v in sd_src.items()
          if not k.startswith('caption_encoder.')}
    if temp is not None:
        with torch.no_grad():
            sd['temp'].fill_(temp)
    args = (sd, WL, batch['image'], batch['ids_s'], batch['mask_s'], batch['ids_t'], batch['mask_t'], batch['ids_c'],
            batch['mask_c'], batch['idx'], text_heads)
    if lowp:        # True: bf16 storage model; 'fp8': + fp8 operands in the encoder blocks' Linears ('fp8_bf16dw': weight gradients in bf16)
        with O.lowp(fp8=(lowp in ('fp8', 'fp8_bf16dw')), fp8_dw=(lowp == 'fp8')):
            losses = O.retrieval_forward(*args)
            sum(losses).backward()
    else:
        losses = O.retrieval_forward(*args)
        sum(losses).backward()
    return [float(l.detach()) for l in losses], sd


def multi30k_batch(b=4):
    from leccr_amd.synthetic import make_batch
    m = make_batch(b)
    return dict(image=m['image'], ids_s=m['text'][0].input_ids, mask_s=m['text'][0].attention_mask,
                ids_t=m['text'][1].input_ids, mask_t=m['text'][1].attention_mask, ids_c=m['captions'].input_ids,
                mask_c=m['captions'].attention_mask, idx=m['idx'])


@pytest.mark.parametrize('dtype', ['fp32', 'bf16'])
def test_full_geometry_model_vs_oracle_multi30k_shapes(dtype):
    """ViT-B/16@384 + mBERT, 12 + 12 layers, MULTI30K_CFG (4 queries, 3 + 2 + 2 interaction layers), b = 4, L_t = 32,
    L_c = 64 - the bench.py model and batch shape - against oracle.retrieval_forward: losses and EVERY parameter gradient."""
    model, _ = full_geometry_model(dtype)
    model = model.to(DEV).eval()
    batch = multi30k_batch(4)
    torch.set_num_threads(16)
    ref_losses, sd = oracle_step(model.state_dict(), batch, 12)
    sd_noise = oracle_step(model.state_dict(), batch, 12, lowp=True)[1] if dtype == 'bf16' else None
    losses = _forward(model, batch)
    got = [float(l) for l in losses]
    tol = 1e-3 if dtype == 'fp32' else 3e-2
    for a, r in zip(got, ref_losses):
        assert abs(a - r) <= tol * max(1.0, abs(r)), (dtype, got, ref_losses)
    model.arena.zero_grad()
    sum(losses).backward()
    rows = grad_rows(model, sd, sd_noise)
    assert len(rows) > 400
    if dtype == 'fp32':
        gmax = max(r[2] for r in rows)
        bad = [(n, err / rn, cos, rn / gmax) for n, err, rn, cos, *_ in rows if rn >= 1e-6 * gmax and err > 2e-3 * rn]
    else:
        bad = outside_noise_model(rows)
        # the yardstick is bounded too: no tensor's modelled bf16 noise may exceed 40 % of its norm (helpers.NOISE_CEILING)
        assert not noise_above_ceiling(rows), noise_above_ceiling(rows)
        out = os.environ.get('LECCR_NOISE_TABLE')
        if out:
            write_noise_table(rows, out, 'ViT-B/16@384 + mBERT (12 + 12 layers), b = 4, L_t 32, L_c 64, bf16 HIP path vs fp32 oracle; noise = oracle.lowp()')
    assert not bad, (len(bad), bad[:8])


# ----------------------------------------------------------------------------------------- 3. bf16 gradients, tensor by tensor
MID_VIT = dict(embed_dim=512, image_resolution=64, vision_layers=2, vision_width=128, vision_patch_size=16,
               context_length=8, vocab_size=32, transformer_width=64, transformer_heads=1, transformer_layers=1)
MID_BERT = dict(vocab_size=500, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                max_position_embeddings=64)


def mid_model(dtype, temp=0.07, seed=3):
    from leccr_amd.models import RetrievalModel
    from tests.test_model_gpu import TINY_CFG
    torch.manual_seed(seed)
    cfg = dict(TINY_CFG, clip_geometry=MID_VIT, text_config=MID_BERT, compute_dtype=dtype, embed_dim=256, clip_text_tower=False,
               temp=temp)
    m = RetrievalModel(cfg)
    m.init_caption_encoder()
    with torch.no_grad():
        m.queries.normal_(0, 0.5)
    return m.to(DEV).eval()


def bf16_vs_noise_model(temp):
    """-> (loss errors, rows of grad_table with the oracle.lowp noise column) for the bf16 HIP path at MID geometry."""
    from tests.test_model_gpu import _mid_batch
    model = mid_model('bf16', temp)
    batch = _mid_batch()
    ref_losses, sd = oracle_step(model.state_dict(), batch, 2)
    noise_losses, sd_noise = oracle_step(model.state_dict(), batch, 2, lowp=True)
    losses = _forward(model, batch)
    model.arena.zero_grad()
    sum(losses).backward()
    got = [float(l) for l in losses]
    return (got, ref_losses, noise_losses), grad_table(model, sd, sd_noise)


@pytest.mark.parametrize('temp', [0.07, 1.0])
def test_bf16_gradients_stay_within_the_bf16_storage_noise_model(temp):
    """InfoNCE at the reference's temp = 0.07 multiplies 2^-9 feature rounding by 14, and some tensors (the query-summary
    branch, whose max-over-queries argmax can flip) are sensitive at any temperature: WHAT a correct bf16 implementation's
    gradient error is differs by orders of magnitude between tensors.  The yardstick is therefore measured, not assumed: the
    oracle re-run with bf16 storage at the points where the HIP path stores bf16 (oracle.lowp) gives each tensor's own noise
    |g_lowp - g_f32|; the HIP bf16 path's error must stay within 3x that (+ 2e-3 of the tensor) for EVERY tensor - no
    allowance by norm class.  temp = 1 repeats it without the amplification (errors and noise both drop ~14x on the
    feature-dominated tensors, so a bug hidden under the amplified noise shows)."""
    (got, ref, noise), rows = bf16_vs_noise_model(temp)
    for a, r, nz in zip(got, ref, noise):
        assert abs(a - r) <= 3 * abs(nz - r) + 2e-3 * max(1.0, abs(r)), (got, ref, noise)
    bad = outside_noise_model(rows)
    assert len(rows) > 100 and not bad, (len(bad), bad[:8])
    assert not noise_above_ceiling(rows), noise_above_ceiling(rows)
    # and the yardstick itself is not vacuous: at least 3/4 of the tensors have noise under 15 % of their norm
    assert sum(1 for _, _, rn, _, nz, *_ in rows if nz <= 0.15 * rn) >= 0.75 * len(rows)


# ----------------------------------------------------------------------------------------- BASELINE configs[4] geometry
L14_GEOMETRY = dict(embed_dim=768, image_resolution=384, vision_layers=2, vision_width=1024, vision_patch_size=14,
                    context_length=8, vocab_size=32, transformer_width=64, transformer_heads=1, transformer_layers=1)
XLMR_LARGE_2L = dict(vocab_size=250002, hidden_size=1024, num_hidden_layers=2, num_attention_heads=16, intermediate_size=4096,
                     max_position_embeddings=512)


@pytest.mark.parametrize('dtype', ['fp32', 'bf16'])
def test_config_e_geometry_vit_l14_xlmr_large_vs_oracle(dtype):
    """BASELINE configs[4]'s geometry at full width: ViT-L/14 @384 (730 tokens = 27 x 27 patches of 14 + [CLS], the 6 edge pixels
    dropped; width 1024, 16 heads, 588 -> 640-wide padded patch rows, vision_width 768: 8 interaction heads of 96 and the wide
    contrastive backward) + an XLM-R-large shaped BERT (hidden 1024, 16 heads, FFN 4096, vocabulary 250 002), two layers of each,
    b = 2, L_t = 32, L_c = 64, against the oracle: losses and every parameter gradient (bf16: the noise-model rule)."""
    from leccr_amd.models import RetrievalModel
    from leccr_amd.synthetic import MULTI30K_CFG, make_batch
    torch.manual_seed(17)
    model = RetrievalModel(dict(MULTI30K_CFG, clip_geometry=L14_GEOMETRY, text_config=XLMR_LARGE_2L, compute_dtype=dtype,
                                clip_text_tower=False))
    model.init_caption_encoder()
    seeded_fill_(model.named_parameters(), 91)
    assert model.vision_width == 768 and model.clip_encoder.visual.positional_embedding.shape == (730, 1024)
    model = model.to(DEV).eval()
    m = make_batch(2, vocab=250002)
    batch = dict(image=m['image'], ids_s=m['text'][0].input_ids, mask_s=m['text'][0].attention_mask, ids_t=m['text'][1].input_ids,
                 mask_t=m['text'][1].attention_mask, ids_c=m['captions'].input_ids, mask_c=m['captions'].attention_mask, idx=m['idx'])
    torch.set_num_threads(16)
    ref_losses, sd = oracle_step(model.state_dict(), batch, 16)
    sd_noise = oracle_step(model.state_dict(), batch, 16, lowp=True)[1] if dtype == 'bf16' else None
    losses = _forward(model, batch)
    tol = 1e-3 if dtype == 'fp32' else 3e-2
    for a, r in zip([float(l) for l in losses], ref_losses):
        assert abs(a - r) <= tol * max(1.0, abs(r)), (dtype, [float(l) for l in losses], ref_losses)
    model.arena.zero_grad()
    sum(losses).backward()
    rows = grad_rows(model, sd, sd_noise)
    assert len(rows) > 150
    if dtype == 'fp32':
        gmax = max(r[2] for r in rows)
        bad = [(n, err / rn, cos, rn / gmax) for n, err, rn, cos, *_ in rows if rn >= 1e-6 * gmax and err > 2e-3 * rn]
    else:
        bad = outside_noise_model(rows)
    assert not bad, (len(bad), bad[:8])


@pytest.mark.parametrize('fuse_ln,dw8', [(False, True), (True, True), (False, False)])
def test_fp8_operand_path_vs_oracle_noise_model(fuse_ln, dw8, monkeypatch):
    _fp8_model_vs_noise_model(fuse_ln, dw8, monkeypatch, L14_GEOMETRY, 8, 2 * 4)


def test_fp8_end_to_end_copies_without_the_bf16_activation(monkeypatch):
    """The same rule where EVERY operand chain is fp8 end to end (round 3): 98 x 98 images (50 tokens) at b = 64 give 3200 token rows = 25
    whole K-tiles of 128 for the ViT blocks and 2048 for the text layers, so from the second step on c_fc / intermediate.dense write
    only the fp8 copy of their activation (no bf16 C), the dX products with the derivative epilogue write only the e5m2 copy of the
    activation gradient and sum c_fc's bias gradient in that epilogue, and all 16 weight-gradient products run from fp8 copies."""
    from leccr_amd import kernels as K
    calls = []
    real = K.gemm

    def spy(*a, **kw):
        calls.append(bool(kw.get('write_c', True)))
        return real(*a, **kw)
    monkeypatch.setattr(K, 'gemm', spy)
    _fp8_model_vs_noise_model(True, True, monkeypatch, dict(L14_GEOMETRY, image_resolution=98), 64, 2 * 4 + 2 * 4)
    # step 2: activation + its gradient in two ViT blocks and two text layers (8) + the activations of the caption encoder's two layers
    # (forward only; it IS the text encoder, so in step 1 its pass already finds the sites the text pass primed: 2 + 2)
    assert calls.count(False) == 2 * (2 + 2) + 2 + 2


def test_fp8_bias_gradients_from_the_e5m2_copy_against_the_unquantised_sums(monkeypatch):
    """Round 4 made the fp8 path take every block Linear's bias gradient from the column sums of the e5m2 (2 mantissa bits) copy of dY inside the
    weight-gradient launch (`ops.fp8_bias_asum`; with it the attention backward and the blocks' LayerNorm passes stop writing the bf16
    tensors).  The accuracy gate the round-4 advisor asked for: the SAME model and batch with the switch on and off
      * every encoder bias gradient of the steady-state (second) step: relative L2 difference between the two settings, and each setting's
        error against the fp32 oracle;
      * six optimizer steps from the same start: the two loss curves stay within 2 % of each other.
    The gate applies to whichever setting is the DEFAULT: the e5m2 sums may be the default only if no bias moves by more than 25 % of its norm
    and none ends more than 2 x (+ 2 %) further from the oracle than with the sums of the unquantised dY.  Measured (round 5, this test's table
    in profiles/r05_fp8_bias_asum_ab.txt): out_proj / c_fc / c_proj biases move by 0.5 - 0.7 of their norm, error 0.43 -> 0.82 at worst - the
    switch went back to opt-in; with the unquantised sums as the default the test holds them to "never further from the oracle than the e5m2
    sums" instead."""
    from leccr_amd import ops
    from leccr_amd.models import RetrievalModel
    from leccr_amd.optim import create_optimizer
    from leccr_amd.scheduler import create_scheduler
    from leccr_amd.synthetic import MULTI30K_CFG, make_batch
    geometry, nb = L14_GEOMETRY, 8
    default_asum = bool(ops.fp8_bias_asum)
    m = make_batch(nb, vocab=30000, res=geometry['image_resolution'])
    batch = dict(image=m['image'], ids_s=m['text'][0].input_ids, mask_s=m['text'][0].attention_mask, ids_t=m['text'][1].input_ids,
                 mask_t=m['text'][1].attention_mask, ids_c=m['captions'].input_ids, mask_c=m['captions'].attention_mask, idx=m['idx'])

    def fresh():
        torch.manual_seed(17)
        model = RetrievalModel(dict(MULTI30K_CFG, clip_geometry=geometry, text_config=dict(XLMR_LARGE_2L, vocab_size=30000),
                                    compute_dtype='fp8', clip_text_tower=False))
        model.init_caption_encoder()
        seeded_fill_(model.named_parameters(), 93)
        return model.to(DEV).eval()

    torch.set_num_threads(16)
    grads, curves = {}, {}
    sd = None
    for asum in (True, False):
        monkeypatch.setattr(ops, 'fp8_bias_asum', asum)
        model = fresh()
        if sd is None:
            _, sd = oracle_step(model.state_dict(), batch, 16)
        for _ in range(2):                                   # the second step is the steady state (producer-written copies, delayed scales)
            losses = _forward(model, batch)
            model.arena.zero_grad()
            sum(losses).backward()
        grads[asum] = {n: p.grad.detach().float().cpu().clone() for n, p in model.named_parameters()
                       if p.grad is not None and n.endswith('bias') and ('resblocks' in n or 'encoder.layer' in n)}
        del model
        model = fresh().train()
        cfg = dict(MULTI30K_CFG)
        opt = create_optimizer(dict(cfg['optimizer'], lr=2e-5), model)
        sched = create_scheduler(dict(cfg['schedular'], step_per_epoch=2, epochs=10), opt)
        curve = []
        for _ in range(6):
            losses = _forward(model, batch)
            opt.zero_grad()
            sum(losses).backward()
            opt.step()
            sched.step()
            curve.append(float(sum(losses).detach()))
        curves[asum] = curve
        del model, opt
    assert len(grads[True]) >= 2 * 6 + 2 * 6 and grads[True].keys() == grads[False].keys()
    table, bad = [], []
    for n, g1 in grads[True].items():
        g0, ref = grads[False][n], sd[n].grad.float()
        rn = float(ref.norm()) + 1e-30
        d10, e1, e0 = float((g1 - g0).norm()) / rn, float((g1 - ref).norm()) / rn, float((g0 - ref).norm()) / rn
        table.append((n, rn, d10, e1, e0))
        if default_asum and (d10 > 0.25 or e1 > 2 * e0 + 0.02):
            bad.append((n, d10, e1, e0))
        if not default_asum and e0 > e1 + 0.02:
            bad.append((n, d10, e1, e0))
    os.makedirs(os.path.join(os.path.dirname(__file__), '..', 'gpurun_out'), exist_ok=True)
    with open(os.path.join(os.path.dirname(__file__), '..', 'gpurun_out', 'fp8_bias_asum_ab.txt'), 'w') as fh:
        fh.write('# bias gradients, second step, ViT-L/14 + XLM-R-large shaped (2 + 2 layers), batch 8: e5m2-copy sums (LECCR_FP8_BIAS_ASUM=1) vs sums of the unquantised dY (0)\n')
        fh.write('# tensor | norm (oracle) | |g1 - g0| / norm | |g1 - oracle| / norm | |g0 - oracle| / norm\n')
        for row in table:
            fh.write('%-70s %.3e  %.4f  %.4f  %.4f\n' % row)
        fh.write('# summed losses over six optimizer steps: asum on  ' + ' '.join('%.5f' % x for x in curves[True]) + '\n')
        fh.write('#                                          asum off ' + ' '.join('%.5f' % x for x in curves[False]) + '\n')
    assert not bad, (len(bad), bad[:6])
    for a, b_ in zip(curves[True], curves[False]):
        assert abs(a - b_) <= 2e-2 * max(1.0, abs(b_)), (curves[True], curves[False])


def _fp8_model_vs_noise_model(fuse_ln, dw8, monkeypatch, geometry, nb, dw_launches):
    """compute_dtype 'fp8' (BASELINE configs[4]: "fp8 MFMA"): the ViT-L/14 + XLM-R-large shaped model (two layers each, b = 8 so
    that every encoder Linear has >= 256 rows and takes the fp8 kernel) with e4m3 activations / weights in the forward products and
    e5m2 gradients in the dX products - and (dw8, round 3) the SAME copies in the weight-gradient products dY^T X of the ViT blocks
    (5840 token rows = 45 K-tiles of 128 on the fp8 kernel + an 80-row bf16 tail; the text layers' 256-row products stay bf16).  Held against the fp32 oracle through the oracle's OWN fp8 operand model (oracle.lowp(fp8=
    True): the same per-tensor power-of-two scaling and formats at the same Linears): losses within 3x what that model moves them,
    every parameter gradient within 3x its noise."""
    from leccr_amd import _lib, ops
    from leccr_amd.models import RetrievalModel
    from leccr_amd.synthetic import MULTI30K_CFG, make_batch
    monkeypatch.setattr(ops, 'fuse_fp8_copies', fuse_ln)      # LECCR_FP8_FUSE_LN: fp8 copies out of the LayerNorm passes (step 2 on)
    monkeypatch.setattr(ops, 'fp8_dw', dw8)                   # LECCR_FP8_DW
    monkeypatch.setattr(ops, 'fp8_dw_launches', 0)
    torch.manual_seed(17)
    model = RetrievalModel(dict(MULTI30K_CFG, clip_geometry=geometry, text_config=dict(XLMR_LARGE_2L, vocab_size=30000),
                                compute_dtype='fp8', clip_text_tower=False))
    model.init_caption_encoder()
    seeded_fill_(model.named_parameters(), 93)
    model = model.to(DEV).eval()
    m = make_batch(nb, vocab=30000, res=geometry['image_resolution'])
    batch = dict(image=m['image'], ids_s=m['text'][0].input_ids, mask_s=m['text'][0].attention_mask, ids_t=m['text'][1].input_ids,
                 mask_t=m['text'][1].attention_mask, ids_c=m['captions'].input_ids, mask_c=m['captions'].attention_mask, idx=m['idx'])
    torch.set_num_threads(16)
    ref_losses, sd = oracle_step(model.state_dict(), batch, 16)
    noise_losses, sd_noise = oracle_step(model.state_dict(), batch, 16, lowp='fp8' if dw8 else 'fp8_bf16dw')
    losses = _forward(model, batch)
    # the fp8 kernel really ran: every encoder Linear's activation site was used, and the last product of the text tower was fp8
    blk = model.clip_encoder.visual.transformer.resblocks[0]
    lay = model.text_encoder.encoder.layer[1]
    assert model.fp8_sites.n == 2 * (2 * 4 + 2 * 4) and model.fp8_weight_sites.n == 2 * 4 + 2 * 4
    assert all(h._xsite.primed for h in (blk.attn.in_proj_weight, blk.mlp.c_fc.weight, blk.mlp.c_proj.weight, lay._qkv_w, lay.output.dense.weight))
    got = [float(l) for l in losses]
    for a, r, nz in zip(got, ref_losses, noise_losses):
        assert abs(a - r) <= 3 * abs(nz - r) + 1e-2 * max(1.0, abs(r)), (got, ref_losses, noise_losses)
    model.arena.zero_grad()
    sum(losses).backward()
    assert all(h._gsite.primed for h in (blk.attn.in_proj_weight, blk.mlp.c_proj.weight, lay._qkv_w, lay.intermediate.dense.weight))
    assert ops.fp8_dw_launches == (dw_launches if dw8 else 0)       # the four Linears of both ViT blocks (+ of both text layers at 2048 rows)
    rows = grad_rows(model, sd, sd_noise)
    bad = outside_noise_model(rows)
    assert len(rows) > 150 and not bad, (len(bad), bad[:8])
    # a second step re-uses the delayed scales (update at the start of forward).  From this step on the producers write the fp8 copies
    # of their outputs themselves - GEMM epilogues (activation / its gradient) and LayerNorm passes (activations forward, gradients
    # backward, the ViT blocks' hand-off included) - instead of separate quantise passes: the steady state of training, held to the
    # same rule.  (Gradients do NOT repeat step 1's: a producer quantises its f32 result, the quantise pass the bf16-rounded tensor,
    # and at fp8 noise levels the lower blocks' gradients are that sensitive - which is what the noise model measures.)
    losses2 = _forward(model, batch)
    got2 = [float(l) for l in losses2]
    for a, r, nz in zip(got2, ref_losses, noise_losses):
        assert abs(a - r) <= 3 * abs(nz - r) + 1e-2 * max(1.0, abs(r)), (got2, ref_losses, noise_losses)
    model.arena.zero_grad()
    sum(losses2).backward()
    rows2 = grad_rows(model, sd, sd_noise)
    bad2 = outside_noise_model(rows2)
    assert len(rows2) > 150 and not bad2, (len(bad2), bad2[:8])


def test_persistent_gemm_paths_in_a_model_vs_oracle():
    """The kernels the benchmark spends its time in, inside a model: ViT-B/16 @384 (one layer) with b = 8 gives 4616 token rows,
    enough for the persistent 256x256 GEMM in BOTH forms - NT forward / dX (>= 192 tiles... the wide ones) and the split-K TN
    weight-gradient product with the bias gradient summed on the matrix cores (`asum`, K = rows >= 4096) - which the b = 2..4
    parity runs above never reach.  bf16, every parameter gradient (weights AND biases of the ViT block) under the noise-model rule."""
    from leccr_amd import _lib, kernels as K
    model, _ = full_geometry_model('bf16', vision_layers=1, text_config=MID_BERT)
    model = model.to(DEV).eval()
    from leccr_amd.synthetic import make_batch
    m = make_batch(8, vocab=500)
    batch = dict(image=m['image'], ids_s=m['text'][0].input_ids, mask_s=m['text'][0].attention_mask, ids_t=m['text'][1].input_ids,
                 mask_t=m['text'][1].attention_mask, ids_c=m['captions'].input_ids, mask_c=m['captions'].attention_mask, idx=m['idx'])
    torch.set_num_threads(16)
    ref_losses, sd = oracle_step(model.state_dict(), batch, 2)
    _, sd_noise = oracle_step(model.state_dict(), batch, 2, lowp=True)
    losses = _forward(model, batch)
    for a, r in zip([float(l) for l in losses], ref_losses):
        assert abs(a - r) <= 3e-2 * max(1.0, abs(r)), ([float(l) for l in losses], ref_losses)
    model.arena.zero_grad()
    lib = _lib.lib()
    lib.leccr_prof_enable(1)
    sum(losses).backward()
    torch.cuda.synchronize()
    lib.leccr_prof_enable(0)
    # the dW products of the ViT block really were the 256x256 TN kernel: a probe launch of the same shape reports kernel 2
    blk = model.clip_encoder.visual.transformer.resblocks[0]
    dy, x = torch.zeros(4616, 3072, device=DEV, dtype=torch.bfloat16), torch.zeros(4616, 768, device=DEV, dtype=torch.bfloat16)
    K.gemm(dy, x, trans_a=True, trans_b=True, out=torch.zeros(3072, 768, device=DEV), accumulate=True, asum=torch.zeros(3072, device=DEV))
    assert lib.leccr_last_gemm_kernel() == 2
    rows = grad_rows(model, sd, sd_noise)
    names = {r[0] for r in rows}
    assert {'clip_encoder.visual.transformer.resblocks.0.mlp.c_fc.bias', 'clip_encoder.visual.transformer.resblocks.0.attn.in_proj_bias',
            'clip_encoder.visual.transformer.resblocks.0.mlp.c_proj.bias', 'clip_encoder.visual.transformer.resblocks.0.attn.out_proj.bias'} <= names
    bad = outside_noise_model(rows)
    assert not bad, (len(bad), bad[:8])
    assert blk.mlp.c_fc.bias.grad.abs().sum() > 0


def test_fp32_attention_and_layernorm_written_copies_change_nothing(monkeypatch):
    """fp32 path, ViT-B/16 @384 (one layer, b = 8: 4616 token rows, head size 64): with the operand copies written by their producers - the
    LayerNorm passes (h1 / h2), the attention kernels (o, the gradient of the packed qkv), one shared copy per operand in the Linears over all
    image tokens outside the blocks (ln_post + proj, the caption interaction, vproj) - every loss and every parameter gradient equals
    the run in which each product splits its own f32 operand, to the pair form's own precision (the backward's delta reads o as hi + lo)."""
    from leccr_amd import kernels as K, ops
    from leccr_amd.synthetic import make_batch
    model, _ = full_geometry_model('fp32', vision_layers=1, text_config=MID_BERT)
    model = model.to(DEV).eval()
    m = make_batch(8, vocab=500)
    batch = dict(image=m['image'], ids_s=m['text'][0].input_ids, mask_s=m['text'][0].attention_mask, ids_t=m['text'][1].input_ids,
                 mask_t=m['text'][1].attention_mask, ids_c=m['captions'].input_ids, mask_c=m['captions'].attention_mask, idx=m['idx'])
    calls = {'fwd': 0, 'bwd': 0}
    fwd_s3, bwd_s3 = K.attn_fwd_s3, K.attn_bwd_s3
    monkeypatch.setattr(K, 'attn_fwd_s3', lambda *a, **k: (calls.__setitem__('fwd', calls['fwd'] + 1), fwd_s3(*a, **k))[1])
    monkeypatch.setattr(K, 'attn_bwd_s3', lambda *a, **k: (calls.__setitem__('bwd', calls['bwd'] + 1), bwd_s3(*a, **k))[1])

    def run(on):
        monkeypatch.setattr(ops, 'attn_x3_copies', on)
        monkeypatch.setattr(ops, 'ln_x3_copies', on)
        monkeypatch.setattr(ops, 'share_x3_copies', on)         # .. and one copy per operand in the Linears outside the blocks
        losses = _forward(model, batch)
        model.arena.zero_grad()
        sum(losses).backward()
        torch.cuda.synchronize()
        return [float(l) for l in losses], {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
    l1, g1 = run(True)
    assert calls == {'fwd': 1, 'bwd': 1}                       # the one ViT block took the attention kernels' own copies
    l0, g0 = run(False)
    assert calls == {'fwd': 1, 'bwd': 1}
    assert max(abs(a - b) for a, b in zip(l1, l0)) <= 1e-5
    # two evaluation orders of the same pair-form arithmetic (2^-17 per product term; the bias sums taken from hi + lo instead of the f32
    # tensor): measured worst 8e-5 of a tensor's largest gradient (class_embedding, out_proj.bias), held to 2e-4.  The packed in-projection
    # biases get 1e-3: their key third is a sum that cancels exactly in exact arithmetic (softmax does not see a constant added to every
    # key), i.e. pure rounding noise of whichever order summed it - measured 3.1e-4 of the tensor's largest entry.
    worst = sorted(((float((g1[k] - g0[k]).abs().max()) / (float(g0[k].abs().max()) + 1e-12), k) for k in g0), reverse=True)
    bad = [(r, k) for r, k in worst if r > (1e-3 if k.endswith('in_proj_bias') else 2e-4)]
    assert len(g0) > 100 and not bad, (bad[:5], worst[:5])


def test_fp32_split_operand_step_does_not_depend_on_the_fp8_weight_gradient_switch(monkeypatch):
    """`LECCR_FP8_DW=0` (ops.fp8_dw: the fp8 path's weight gradients back on the bf16 kernel) is an fp8 switch - but the fp32 path hands its bf16
    split copies through the same hand-over points, and until round 5 BertLayerFn / ViTBlockFn saved those copies only `if fp8_dw`, AFTER
    dw_keep() had dropped the f32 tensors they stand for: the fp32 step died in lin_dw ("a split copy without its partner"), out_proj's bias
    gradient counted twice on the way (round-4 advisor finding).  The same model and batch with the switch on and off: identical losses and
    gradients up to the order of the split-K atomics (the switch must not reach the fp32 path at all)."""
    from leccr_amd import ops
    from leccr_amd.synthetic import make_batch
    model, _ = full_geometry_model('fp32', vision_layers=1, text_config=MID_BERT)
    model = model.to(DEV).eval()
    m = make_batch(8, vocab=500)
    batch = dict(image=m['image'], ids_s=m['text'][0].input_ids, mask_s=m['text'][0].attention_mask, ids_t=m['text'][1].input_ids,
                 mask_t=m['text'][1].attention_mask, ids_c=m['captions'].input_ids, mask_c=m['captions'].attention_mask, idx=m['idx'])

    def run(dw8):
        monkeypatch.setattr(ops, 'fp8_dw', dw8)
        losses = _forward(model, batch)
        model.arena.zero_grad()
        sum(losses).backward()
        torch.cuda.synchronize()
        return [float(l) for l in losses], {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
    l1, g1 = run(True)
    l0, g0 = run(False)
    assert max(abs(a - b) for a, b in zip(l1, l0)) <= 1e-6
    # (the weight gradients are split-K sums of f32 atomics: the order of the adds differs from launch to launch)
    worst = sorted(((float((g1[k] - g0[k]).abs().max()) / (float(g0[k].abs().max()) + 1e-12), k) for k in g0), reverse=True)
    assert len(g0) > 100 and worst[0][0] <= 2e-4, worst[:5]           # (measured 5.7e-5 on ln_post.bias: two runs of the same code differ by that)


def test_fp8_weight_copies_and_their_scale_stay_together_when_a_weight_grows():
    """A weight's fp8 copies are written once per optimizer step and read by every product until the next one, while the 1/scale of
    the product is read at kernel time: both must come from the SAME scale.  Round 2 requantised under the site's previous scale and
    let the per-step Fp8Sites.update() replace it afterwards - silently 2x / 0.5x off for one step whenever a weight's amax crossed a
    power of two.  Here two Linears' weights are multiplied by 4 and by 1/4 between two steps (what many optimizer steps do slowly);
    the fp8 tower must stay as close to the bf16 tower with the same weights as it was before."""
    from leccr_amd.models import RetrievalModel
    from leccr_amd.synthetic import MULTI30K_CFG, make_batch

    def tower(dtype):
        torch.manual_seed(17)
        m = RetrievalModel(dict(MULTI30K_CFG, clip_geometry=L14_GEOMETRY, text_config=dict(XLMR_LARGE_2L, vocab_size=30000),
                                compute_dtype=dtype, clip_text_tower=False))
        m.init_caption_encoder()
        seeded_fill_(m.named_parameters(), 93)
        return m.to(DEV).eval()
    m8, m16 = tower('fp8'), tower('bf16')
    mb = make_batch(8, vocab=30000)
    batch = dict(image=mb['image'], ids_s=mb['text'][0].input_ids, mask_s=mb['text'][0].attention_mask, ids_t=mb['text'][1].input_ids,
                 mask_t=mb['text'][1].attention_mask, ids_c=mb['captions'].input_ids, mask_c=mb['captions'].attention_mask, idx=mb['idx'])
    image = batch['image'].to(DEV)

    def rel(a, b):
        return float((a.float() - b.float()).norm() / b.float().norm())
    with torch.no_grad():
        _forward(m8, batch)                                   # step 1: primes the activation sites
        before = rel(m8.get_vision_embeds(image)[0], m16.get_vision_embeds(image)[0])
        for m in (m8, m16):
            blk = m.clip_encoder.visual.transformer.resblocks[0]
            blk.mlp.c_fc.weight.mul_(4.0)
            blk.attn.out_proj.weight.mul_(0.25)
            m.refresh_weights()                               # what optimizer.step() calls: compute copies + fp8 requantisation
        _forward(m8, batch)                                   # step 2: Fp8Sites.update() runs first - it must not move a weight's scale
        after = rel(m8.get_vision_embeds(image)[0], m16.get_vision_embeds(image)[0])
        # the copies and the scale the products read belong together: dequantised, they ARE the bf16 weights to e4m3 rounding (2^-4
        # relative per element; the round-2 bug left them 4x / 0.25x apart for a step)
        blk = m8.clip_encoder.visual.transformer.resblocks[0]
        for h in (blk.mlp.c_fc.weight, blk.attn.out_proj.weight, blk.attn.in_proj_weight):
            inv = float(h._wsite.inv_scale)
            for q8, lp in ((h._q8, h._lp), (h._q8t, h._lpt)):
                deq = q8.float() * inv
                assert rel(deq, lp) < 0.04, (rel(deq, lp), inv)
                assert float(deq.abs().max()) <= 1.07 * float(lp.float().abs().max())
    assert before < 0.1, before
    # (the 4x larger MLP branch of block 0 carries 4x its fp8 noise into the residual stream: measured 0.053 -> 0.138)
    assert after < 0.3, (before, after)
