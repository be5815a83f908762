import os, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch
from tests.test_full_geometry_gpu import *
from tests.test_full_geometry_gpu import _forward
from leccr_amd.models import RetrievalModel
from leccr_amd.synthetic import MULTI30K_CFG, make_batch
torch.manual_seed(17)
model = RetrievalModel(dict(MULTI30K_CFG, clip_geometry=L14_GEOMETRY, text_config=dict(XLMR_LARGE_2L, vocab_size=30000), compute_dtype='fp8', clip_text_tower=False))
model.init_caption_encoder()
seeded_fill_(model.named_parameters(), 93)
model = model.to(DEV).eval()
m = make_batch(8, vocab=30000)
batch = dict(image=m['image'], ids_s=m['text'][0].input_ids, mask_s=m['text'][0].attention_mask, ids_t=m['text'][1].input_ids,
             mask_t=m['text'][1].attention_mask, ids_c=m['captions'].input_ids, mask_c=m['captions'].attention_mask, idx=m['idx'])
gs = []
for step in range(3):
    losses = _forward(model, batch)
    model.arena.zero_grad()
    sum(losses).backward()
    gs.append({n: p._g.detach().clone() for n, p in model.named_parameters() if getattr(p, '_g', None) is not None})
    print('step', step, [round(float(l), 5) for l in losses])
for a, b in ((0, 1), (1, 2)):
    d = sorted(((float((gs[b][n] - gs[a][n]).norm() / (gs[a][n].norm() + 1e-12)), n, float(gs[a][n].norm())) for n in gs[a]), reverse=True)
    print('steps', a, b, 'worst:', [(round(x, 4), n, f'{nn:.2e}') for x, n, nn in d[:6]])
