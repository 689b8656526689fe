"""PatchGAN discriminator on libwfae.so kernels.

Mirrors `pipeline/models/autoencoderkl/losses/model.py` of the reference: `weights_init` (:6-12) and
`NLayerDiscriminator` (:100-150) with the same constructor signature, `self.main` Sequential and
state_dict keys (`main.{0,2,3,5,6,8,9,11}.*`), so checkpoints interchange.  forward() walks `main` and
runs each Conv -> BatchNorm -> LeakyReLU triple as one fused autograd Function.
"""
from __future__ import annotations

import torch
import torch.nn as tnn

from ..... import functional as Fn
from ..... import nn as wnn
from ....._lib import WfaeError


def weights_init(m):
    """N(0, 0.02) conv weights, N(1, 0.02) BatchNorm gamma, zero beta (reference model.py:6-12)."""
    name = m.__class__.__name__
    if name.find("Conv") != -1:
        tnn.init.normal_(m.weight.data, 0.0, 0.02)
    elif name.find("BatchNorm") != -1:
        tnn.init.normal_(m.weight.data, 1.0, 0.02)
        tnn.init.constant_(m.bias.data, 0)


class NLayerDiscriminator(tnn.Module):
    def __init__(self, input_nc=3, ndf=64, n_layers=3, use_actnorm=False):
        super().__init__()
        if use_actnorm:
            raise WfaeError("NLayerDiscriminator: ActNorm is not built (use_actnorm is false in every shipped config)")
        seq = [wnn.Conv2d(input_nc, ndf, kernel_size=4, stride=2, padding=1), wnn.LeakyReLU(0.2, True)]
        mult = 1
        for n in range(1, n_layers):
            prev, mult = mult, min(2 ** n, 8)
            seq += [wnn.Conv2d(ndf * prev, ndf * mult, kernel_size=4, stride=2, padding=1, bias=False),
                    wnn.BatchNorm2d(ndf * mult), wnn.LeakyReLU(0.2, True)]
        prev, mult = mult, min(2 ** n_layers, 8)
        seq += [wnn.Conv2d(ndf * prev, ndf * mult, kernel_size=4, stride=1, padding=1, bias=False),
                wnn.BatchNorm2d(ndf * mult), wnn.LeakyReLU(0.2, True)]
        # the reference's output layer: a 1x1 convolution with padding=1 (a zero-padded border whose
        # logits equal the bias) — kept as is (model.py:144-145)
        seq += [wnn.Conv2d(ndf * mult, 1, kernel_size=1, stride=1, padding=1)]
        self.main = tnn.Sequential(*seq)

    def forward(self, input):
        x = input
        mods = list(self.main)
        i = 0
        while i < len(mods):
            m = mods[i]
            nxt = mods[i + 1] if i + 1 < len(mods) else None
            nxt2 = mods[i + 2] if i + 2 < len(mods) else None
            if isinstance(m, wnn.Conv2d) and m.kernel_size == (4, 4) and m.padding == (1, 1):
                if isinstance(nxt, wnn.BatchNorm2d) and isinstance(nxt2, wnn.LeakyReLU) and m.bias is None:
                    x = Fn.DiscUnitFn.apply(x, m.weight, nxt.weight, nxt.bias, nxt, m.stride[0])
                    i += 3
                    continue
                if isinstance(nxt, wnn.LeakyReLU) and m.stride == (2, 2):
                    x = Fn.Conv4LeakyFn.apply(x, m.weight, m.bias)
                    i += 2
                    continue
            x = m(x)
            i += 1
        return x
