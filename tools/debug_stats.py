"""Accuracy of the two BatchNorm-statistics paths against fp64 on the same stored tensor."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from weatherforecastingtoolkit_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
for nb, cin, cout, h in [(1, 64, 256, 192), (4, 64, 256, 192), (1, 256, 64, 192), (8, 256, 64, 96)]:
    x = torch.randn(nb, cin, h, h, device=dev)
    w = torch.randn(cout, cin, 1, 1, device=dev) * cin ** -0.5
    r = torch.randn(nb, cout, h, h, device=dev) + 0.7
    gamma, beta = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
    y, sr = ops.conv1x1_fwd_stats(x, w, None, r)
    yd = y.double()
    mean = yd.mean(dim=(0, 2, 3))
    var = yd.var(dim=(0, 2, 3), unbiased=False)
    inv = (var + 1e-5).rsqrt()
    a = ops.bn_stats_train(y, gamma, beta, torch.zeros(cout, device=dev), torch.ones(cout, device=dev))
    line = f"{nb}x{cout}@{h}: standalone mean {float(((a.mean.double() - mean).abs() / var.sqrt()).max()):.2e} invstd {float(((a.invstd.double() - inv).abs() / inv).max()):.2e}"
    if sr is not None:
        b = ops.bn_stats_from_rows(sr, tuple(y.shape), gamma, beta, torch.zeros(cout, device=dev), torch.ones(cout, device=dev))
        line += f" | fused mean {float(((b.mean.double() - mean).abs() / var.sqrt()).max()):.2e} invstd {float(((b.invstd.double() - inv).abs() / inv).max()):.2e} rows {sr.rows}"
        s = sr.part[:sr.rows * cout].view(sr.rows, cout).double().sum(0)
        line += f" | raw sum err {float(((s / (nb * h * h) - mean).abs() / var.sqrt()).max()):.2e}"
    else:
        line += " | not served"
    print(line)
