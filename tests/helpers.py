"""Shared helpers for the tests: fixture loading and comparison."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load_npz(name):
    return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


def weights(npz, prefix='w/', requires_grad=False, device='cpu'):
    sd = {}
    for k, v in npz.items():
        if k.startswith(prefix):
            t = torch.from_numpy(np.array(v)).to(device)
            if requires_grad and t.is_floating_point():
                t.requires_grad_(True)
            sd[k[len(prefix):]] = t
    return sd


def T(a, device='cpu'):
    return torch.from_numpy(np.array(a)).to(device)


def max_err(a, b):
    a = a.detach().cpu().double().numpy() if torch.is_tensor(a) else np.asarray(a, dtype=np.float64)
    b = b.detach().cpu().double().numpy() if torch.is_tensor(b) else np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.max(np.abs(a - b))) if a.size else 0.0


def rel_err(a, b):
    """max |a-b| / max|b| (scale-aware, for gradients whose magnitude varies by tensor)."""
    b_ = b.detach().cpu().double().numpy() if torch.is_tensor(b) else np.asarray(b, dtype=np.float64)
    return max_err(a, b) / (float(np.max(np.abs(b_))) + 1e-12)


def grad_close(a, b, rtol, atol=1e-6):
    """|a-b| <= atol + rtol*max|b|  (mathematically-zero gradients, e.g. BERT key biases, are ~1e-8 noise)."""
    b_ = b.detach().cpu().double().numpy() if torch.is_tensor(b) else np.asarray(b, dtype=np.float64)
    return max_err(a, b) <= atol + rtol * float(np.max(np.abs(b_)))
