"""Uninitialised-read hunt: torch.empty filled with NaN / max-int (torch.utils.deterministic.fill_uninitialized_memory) under the fp8 / bf16 / fp32
model step of tests/test_full_geometry_gpu.py's config-E shaped model - any kernel that reads memory nobody wrote shows up as NaN in losses / gradients."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.use_deterministic_algorithms(True, warn_only=True)
torch.utils.deterministic.fill_uninitialized_memory = True
from tests.test_full_geometry_gpu import L14_GEOMETRY, XLMR_LARGE_2L, _forward, seeded_fill_, DEV
from leccr_amd import ops
from leccr_amd.models import RetrievalModel
from leccr_amd.synthetic import MULTI30K_CFG, make_batch
dtype = sys.argv[1] if len(sys.argv) > 1 else 'fp8'
res = int(sys.argv[2]) if len(sys.argv) > 2 else L14_GEOMETRY['image_resolution']
nb = int(sys.argv[3]) if len(sys.argv) > 3 else 8
ops.fp8_bias_asum = os.environ.get('ASUM', '0') == '1'
geometry = dict(L14_GEOMETRY, image_resolution=res)
m = make_batch(nb, vocab=30000, res=res)
batch = dict(image=m['image'], ids_s=m['text'][0].input_ids, mask_s=m['text'][0].attention_mask, ids_t=m['text'][1].input_ids,
             mask_t=m['text'][1].attention_mask, ids_c=m['captions'].input_ids, mask_c=m['captions'].attention_mask, idx=m['idx'])
torch.manual_seed(17)
model = RetrievalModel(dict(MULTI30K_CFG, clip_geometry=geometry, text_config=dict(XLMR_LARGE_2L, vocab_size=30000), compute_dtype=dtype, clip_text_tower=False))
model.init_caption_encoder()
seeded_fill_(model.named_parameters(), 93)
model = model.to(DEV).eval()
for step in range(3):
    losses = _forward(model, batch)
    model.arena.zero_grad()
    sum(losses).backward()
    torch.cuda.synchronize()
    bad = [n for n, p in model.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
    print(dtype, res, nb, 'asum', ops.fp8_bias_asum, 'step', step, 'losses', [round(float(l), 5) for l in losses], 'non-finite gradients:', len(bad), bad[:6], flush=True)
