"""Diagnostic: per-parameter gradient agreement between the bf16 HIP path and the f32 CPU oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.test_model_gpu import build, _mid_batch, _oracle_step, MID_VIT, MID_BERT, Enc, DEV

for dtype in ('fp32', 'bf16'):
    model = build(dict(clip_geometry=MID_VIT, text_config=MID_BERT, compute_dtype=dtype, embed_dim=256, clip_text_tower=False), seed=3)
    with torch.no_grad():
        model.queries.normal_(0, 0.5)
    batch = _mid_batch()
    ref_losses, sd = _oracle_step(model, batch, 2)
    losses = model(batch['image'].to(DEV), [Enc(batch['ids_s'].to(DEV), batch['mask_s'].to(DEV)), Enc(batch['ids_t'].to(DEV), batch['mask_t'].to(DEV))],
                   captions=Enc(batch['ids_c'].to(DEV), batch['mask_c'].to(DEV)), idx=batch['idx'].to(DEV))
    print(dtype, 'losses', [round(float(l), 5) for l in losses], 'ref', [round(l, 5) for l in ref_losses])
    model.arena.zero_grad()
    sum(losses).backward()
    rows = []
    for n, p in model.named_parameters():
        r = sd[n].grad
        if r is None:
            continue
        g = p.grad.detach().cpu().double()
        r = r.double()
        rel = float((g - r).norm() / (r.norm() + 1e-30))
        cos = float((g * r).sum() / (g.norm() * r.norm() + 1e-30))
        rows.append((rel, cos, float(r.norm()), n))
    rows.sort(reverse=True)
    for row in rows[:25]:
        print('  rel %.3e cos %.5f |r| %.3e %s' % row)
