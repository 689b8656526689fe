"""Full model (g4 input) with and without fused BatchNorm sums in one process: outputs of every Bottleneck and all
gradients compared elementwise."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests._util import golden  # noqa: E402
from tests.test_model_gpu import _build, _frames  # noqa: E402
from weatherforecastingtoolkit_amd import functional as Fn  # noqa: E402
from weatherforecastingtoolkit_amd.pipeline.models.ae_64x8x8_lin import Bottleneck  # noqa: E402

g = golden("g4_full384_b1")
dev = torch.device("cuda:0")
net = _build(int(g["img_size"]), dev)
x = _frames(g).to(dev)
rel = lambda u, v: float((u.double() - v.double()).abs().max() / (v.double().abs().max() + 1e-30))
runs = {}
for fuse in (False, True):
    Fn.STAT_FUSION = Fn.PRODUCER_STATS = fuse
    outs = {}
    hooks = [m.register_forward_hook(lambda mod, i, o, n=n: outs.__setitem__(n, o.detach().clone()))
             for n, m in net.named_modules() if isinstance(m, Bottleneck)]
    for p in net.parameters():
        p.grad = None
    for m in net.modules():            # same running statistics at the start of both runs
        if hasattr(m, "running_mean") and m.running_mean is not None:
            m.running_mean.zero_(); m.running_var.fill_(1.0)
    recon, z = net(x)
    loss = Fn.l1_loss(recon, x)
    loss.backward()
    for h in hooks:
        h.remove()
    runs[fuse] = (outs, {n: p.grad.detach().clone() for n, p in net.named_parameters()}, recon.detach().clone())
a, b = runs[False], runs[True]
print("recon", rel(b[2], a[2]))
for n in a[0]:
    print(f"  out {n:14s} {rel(b[0][n], a[0][n]):.2e}")
worst = sorted(((rel(b[1][n], a[1][n]), n) for n in a[1]), reverse=True)[:10]
for r, n in worst:
    print(f"  grad {n:28s} {r:.2e}")
