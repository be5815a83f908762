#!/usr/bin/env python3
"""Print the per-tensor numbers behind the bounds of tests/test_full_geometry_gpu.py (run on the GPU box):
HIP bf16 error vs the oracle's bf16-storage noise model at temp 0.07 and 1.0, and the full-geometry error tables."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import test_full_geometry_gpu as F  # noqa: E402


def show(title, rows, k=12):
    gmax = max(r[2] for r in rows)
    live = [r for r in rows if r[2] >= 1e-6 * gmax]
    print(f'== {title}: {len(live)} tensors')
    worst = sorted(live, key=lambda r: -(r[1] / r[2]))[:k]
    for n, err, rn, cos, nz, *_ in worst:
        print(f'  rel {err / rn:9.3e}  noise {((nz or 0) / rn):9.3e}  ratio {(err / nz if nz else float("nan")):6.2f}  cos {cos:.5f}  |g|/gmax {rn / gmax:8.2e}  {n}')
    if live and live[0][4] is not None:
        worst = sorted(live, key=lambda r: -(r[1] / (3 * r[4] + 1e-3 * r[2])))[:k]
        print('  -- by err / (3 noise + 1e-3 |g|)')
        for n, err, rn, cos, nz, *_ in worst:
            print(f'  score {err / (3 * nz + 1e-3 * rn):6.3f} rel {err / rn:9.3e}  noise {nz / rn:9.3e}  cos {cos:.5f}  {n}')


if __name__ == '__main__':
    for temp in ((0.07, 1.0) if 'mid' in sys.argv else ()):
        (got, ref, noise), rows = F.bf16_vs_noise_model(temp)
        print('temp', temp, 'losses hip', got, '\n   ref', ref, '\n   noise', noise)
        show(f'MID bf16 temp {temp}', rows)
    for dtype in (('bf16',) if 'full' in sys.argv else ()):
        model, _ = F.full_geometry_model(dtype)
        model = model.to('cuda').eval()
        batch = F.multi30k_batch(4)
        torch.set_num_threads(16)
        ref_losses, sd = F.oracle_step(model.state_dict(), batch, 12)
        sd_noise = F.oracle_step(model.state_dict(), batch, 12, lowp=True)[1] if dtype == 'bf16' else None
        losses = F._forward(model, batch)
        model.arena.zero_grad()
        sum(losses).backward()
        print(dtype, 'full losses hip', [float(l) for l in losses], 'ref', ref_losses)
        show(f'FULL {dtype}', F.grad_table(model, sd, sd_noise), k=20)
        del model
        torch.cuda.empty_cache()
