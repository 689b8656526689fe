import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def relerr(a, b):
    """max |a-b| / (max |b| + tiny), both numpy or torch (moved to cpu)."""
    import torch
    if isinstance(a, torch.Tensor):
        a = a.detach().double().cpu().numpy()
    if isinstance(b, torch.Tensor):
        b = b.detach().double().cpu().numpy()
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-30))
