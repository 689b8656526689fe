"""Drop-in for the reference's pipeline/models/ae_64x8x8_lin.py on MI355X.

Same class names, constructor signatures, sub-module names and state_dict
keys/shapes as the reference file
(pipeline/models/ae_64x8x8_lin.py:7-106), so `strict=True` checkpoint loading
works in both directions; forward/backward run on libwfae.so kernels.

Extension over the reference: kw-only `img_size` (default 128, the only size
the reference supports — its bottleneck glue is hard-wired to 8x8, :72-75,100).
`img_size=384` sizes pos_emb / to_latent / from_latent for the 24x24 latent
map; the conv stacks are size-agnostic.
"""
from __future__ import annotations

import torch
import torch.nn as tnn

from ... import functional as Fn
from ... import nn as wnn


class Bottleneck(tnn.Module):
    """Pre-activation bottleneck: x + 1x1(GELU(BN(g3x3(GELU(BN(1x1(GELU(BN(x)))))))))
    — reference ae_64x8x8_lin.py:7-22.  Fused: one autograd node, 13 forward launches."""

    def __init__(self, channels: int, groups: int = 8):
        super().__init__()
        mid = channels // 4
        g = min(groups, mid)
        assert mid % g == 0, f"groups ({g}) must divide mid channels ({mid})"
        self.f = tnn.Sequential(
            wnn.BatchNorm2d(channels), wnn.GELU(),
            wnn.Conv2d(channels, mid, 1, bias=False),
            wnn.BatchNorm2d(mid), wnn.GELU(),
            wnn.Conv2d(mid, mid, 3, padding=1, groups=g, bias=False),
            wnn.BatchNorm2d(mid), wnn.GELU(),
            wnn.Conv2d(mid, channels, 1, bias=False),
        )

    def forward(self, x):
        f = self.f
        return Fn.BottleneckFn.apply(x, f[0].weight, f[0].bias, f[2].weight, f[3].weight, f[3].bias,
                                     f[5].weight, f[6].weight, f[6].bias, f[8].weight, self)


class EncBlock(tnn.Module):
    """Conv2d(4, s2, p1, no bias) -> BN -> GELU -> bottlenecks — reference :27-36."""

    def __init__(self, in_ch: int, out_ch: int, num_blocks: int = 2, groups: int = 8):
        super().__init__()
        self.down = tnn.Sequential(
            wnn.Conv2d(in_ch, out_ch, 4, stride=2, padding=1, bias=False),
            wnn.BatchNorm2d(out_ch), wnn.GELU(),
        )
        self.res = tnn.Sequential(*[Bottleneck(out_ch, groups) for _ in range(num_blocks)])

    def forward(self, x):
        d = self.down
        return self.res(Fn.DownUnitFn.apply(x, d[0].weight, d[1].weight, d[1].bias, d[1]))


class DecBlock(tnn.Module):
    """ConvTranspose2d(4, s2, p1, no bias) -> BN -> GELU -> bottlenecks — reference :38-47."""

    def __init__(self, in_ch: int, out_ch: int, num_blocks: int = 2, groups: int = 8):
        super().__init__()
        self.up = tnn.Sequential(
            wnn.ConvTranspose2d(in_ch, out_ch, 4, stride=2, padding=1, bias=False),
            wnn.BatchNorm2d(out_ch), wnn.GELU(),
        )
        self.res = tnn.Sequential(*[Bottleneck(out_ch, groups) for _ in range(num_blocks)])

    def forward(self, x):
        u = self.up
        return self.res(Fn.UpUnitFn.apply(x, u[0].weight, u[1].weight, u[1].bias, u[1]))


class PosAwareAE_TF(tnn.Module):
    """Conv autoencoder 1xHxW -> latent_dim -> 1xHxW — reference :52-106."""

    def __init__(self, in_channels: int = 1, latent_channels: int = 64, groups: int = 8,
                 latent_dim: int = 2048, *, img_size: int = 128):
        super().__init__()
        assert img_size % 16 == 0
        self.latent_channels = latent_channels
        self.img_size = img_size
        hw = img_size // 16
        self.latent_hw = hw

        self.enc = tnn.Sequential(
            EncBlock(in_channels, 256, num_blocks=4, groups=groups),
            EncBlock(256, 512, num_blocks=4, groups=groups),
            EncBlock(512, 1024, num_blocks=4, groups=groups),
            EncBlock(1024, 1024, num_blocks=4, groups=groups),
            wnn.Conv2d(1024, latent_channels, 1),
        )
        self.pos_emb = tnn.Parameter(torch.randn(1, latent_channels, hw, hw))

        self.to_latent = wnn.Linear(hw * hw * latent_channels, latent_dim)
        self.from_latent = wnn.Linear(latent_dim, hw * hw * latent_channels)

        self.dec = tnn.Sequential(
            wnn.Conv2d(latent_channels, 1024, 1),
            DecBlock(1024, 1024, num_blocks=4, groups=groups),
            DecBlock(1024, 512, num_blocks=4, groups=groups),
            DecBlock(512, 256, num_blocks=4, groups=groups),
            DecBlock(256, 128, num_blocks=4, groups=groups),
            wnn.Conv2d(128, in_channels, 3, padding=1),
        )
        self.act = wnn.Sigmoid()

    def encode(self, x):
        # enc(x) + pos_emb -> flatten -> to_latent (reference :88-94); the
        # broadcast pos_emb add is fused into the epilogue of the enc[4] 1x1 conv.
        h = x
        for blk in list(self.enc)[:-1]:
            h = blk(h)
        last = self.enc[-1]
        z = Fn.Conv1x1Fn.apply(h, last.weight, last.bias, self.pos_emb)
        return self.to_latent(z.flatten(1))

    def decode(self, z_flat):
        B = z_flat.size(0)
        z = self.from_latent(z_flat)
        z = z.view(B, self.latent_channels, self.latent_hw, self.latent_hw)
        return self.act(self.dec(z))

    def forward(self, x):
        z = self.encode(x)
        return self.decode(z), z
