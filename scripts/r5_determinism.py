"""Two identical fp8 steps of the ViT-L/14 + XLM-R-large shaped model (2 + 2 layers, batch 8, eval mode): are the gradients reproducible?
(the weight-gradient products sum with f32 atomics: last-bit differences are expected there; anything larger is a race or an uninitialised read)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.test_full_geometry_gpu import L14_GEOMETRY, XLMR_LARGE_2L, _forward, seeded_fill_, DEV
from leccr_amd.models import RetrievalModel
from leccr_amd.synthetic import MULTI30K_CFG, make_batch
dtype = sys.argv[1] if len(sys.argv) > 1 else 'fp8'
geometry, nb = L14_GEOMETRY, 8
m = make_batch(nb, vocab=30000, res=geometry['image_resolution'])
batch = dict(image=m['image'], ids_s=m['text'][0].input_ids, mask_s=m['text'][0].attention_mask, ids_t=m['text'][1].input_ids,
             mask_t=m['text'][1].attention_mask, ids_c=m['captions'].input_ids, mask_c=m['captions'].attention_mask, idx=m['idx'])
def fresh():
    torch.manual_seed(17)
    model = RetrievalModel(dict(MULTI30K_CFG, clip_geometry=geometry, text_config=dict(XLMR_LARGE_2L, vocab_size=30000), compute_dtype=dtype, clip_text_tower=False))
    model.init_caption_encoder()
    seeded_fill_(model.named_parameters(), 93)
    return model.to(DEV).eval()
from leccr_amd import ops
if os.environ.get('PRELUDE') == '1':       # another fp8 model first (the geometry of test_fp8_end_to_end_copies_without_the_bf16_activation)
    g2 = dict(L14_GEOMETRY, image_resolution=98)
    m2 = make_batch(64, vocab=30000, res=98)
    b2 = dict(image=m2['image'], ids_s=m2['text'][0].input_ids, mask_s=m2['text'][0].attention_mask, ids_t=m2['text'][1].input_ids,
              mask_t=m2['text'][1].attention_mask, ids_c=m2['captions'].input_ids, mask_c=m2['captions'].attention_mask, idx=m2['idx'])
    torch.manual_seed(17)
    pm = RetrievalModel(dict(MULTI30K_CFG, clip_geometry=g2, text_config=dict(XLMR_LARGE_2L, vocab_size=30000), compute_dtype=dtype, clip_text_tower=False))
    pm.init_caption_encoder()
    seeded_fill_(pm.named_parameters(), 93)
    pm = pm.to(DEV).eval()
    for _ in range(2):
        ls = _forward(pm, b2)
        pm.arena.zero_grad()
        sum(ls).backward()
    torch.cuda.synchronize()
    del pm, ls
runs = []
for r in range(2):
    if os.environ.get('ASUM_AB') == '1':
        ops.fp8_bias_asum = r == 0
    model = fresh()
    L = []
    for _ in range(2):
        losses = _forward(model, batch)
        model.arena.zero_grad()
        sum(losses).backward()
        L.append([float(l) for l in losses])
    torch.cuda.synchronize()
    runs.append((L, {n: p.grad.detach().float().cpu().clone() for n, p in model.named_parameters() if p.grad is not None}))
    del model
print('losses equal:', runs[0][0] == runs[1][0], runs[0][0], runs[1][0])
worst = sorted(((float((runs[0][1][n] - runs[1][1][n]).norm() / (runs[0][1][n].norm() + 1e-30)), n) for n in runs[0][1]), reverse=True)
print('ACT_DERIV', os.environ.get('LECCR_ACT_DERIV', '1'), dtype, 'worst relative gradient differences between two identical runs:')
for d, n in worst[:8] + [w for w in worst if 'text_encoder.encoder.layer.0.output.dense' in w[1] or 'resblocks.1.ln_1' in w[1]]:
    print('  %.3e  %s' % (d, n))
