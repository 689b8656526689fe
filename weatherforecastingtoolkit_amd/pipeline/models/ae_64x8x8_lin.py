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


# channel plans of the four encoder / decoder stages and the residual depth per stage (reference :58-62, :76-82)
_ENC_WIDTHS = (256, 512, 1024, 1024)
_DEC_WIDTHS = (1024, 512, 256, 128)
_BLOCKS_PER_STAGE = 4


def _norm_act_conv(c_in, c_out, k, **conv_kw):
    """[BatchNorm2d(c_in), GELU, Conv2d(c_in -> c_out, k, bias=False)] — one third of a Bottleneck body"""
    return [wnn.BatchNorm2d(c_in), wnn.GELU(), wnn.Conv2d(c_in, c_out, k, bias=False, **conv_kw)]


def _residual_stack(width, depth, groups):
    blocks = [Bottleneck(width, groups) for _ in range(depth)]
    for blk in blocks[:-1]:        # the next module is a Bottleneck, whose first BatchNorm wants the sums of this output
        blk.emit_stats = True
    return tnn.Sequential(*blocks)


def _with_unit_stats(a, bn):
    """hand the BatchNorm sums of a unit's output (reduced by the kernel that wrote it, functional.PRODUCER_STATS) to the
    first Bottleneck through the tensor attribute Bottleneck.forward reads"""
    sp, bn._out_stats = getattr(bn, "_out_stats", None), None
    if sp is not None:
        a._wfae_stats = sp
    return a


class Bottleneck(tnn.Module):
    """Pre-activation bottleneck  x + W3 . gelu(bn(G3x3 . gelu(bn(W1 . gelu(bn(x))))))  at a quarter of the width in
    the middle (reference ae_64x8x8_lin.py:7-22; `f.0 … f.8` are its Sequential indices).  Runs as ONE autograd
    node (functional.BottleneckFn) with the residual add fused into the last 1x1 epilogue.  With
    functional.STAT_FUSION (off by default) the BatchNorm sums of both 1x1 outputs also ride in the GEMM epilogues; the
    sums of the block output travel to the next block as the `_wfae_stats` attribute."""

    def __init__(self, channels: int, groups: int = 8):
        super().__init__()
        inner = channels // 4
        n_groups = min(groups, inner)
        assert inner % n_groups == 0, f"groups ({n_groups}) must divide mid channels ({inner})"
        body = (_norm_act_conv(channels, inner, 1)
                + _norm_act_conv(inner, inner, 3, padding=1, groups=n_groups)
                + _norm_act_conv(inner, channels, 1))
        self.f = tnn.Sequential(*body)

    emit_stats = False   # set by the enclosing stack: reduce the BatchNorm sums of the output for the next block
    _out_stats = None

    def forward(self, x):
        bn1, _, w1, bn2, _, wg, bn3, _, w3 = self.f
        y = Fn.BottleneckFn.apply(x, bn1.weight, bn1.bias, w1.weight, bn2.weight, bn2.bias, wg.weight,
                                  bn3.weight, bn3.bias, w3.weight, self, getattr(x, "_wfae_stats", None))
        if self._out_stats is not None:
            y._wfae_stats, self._out_stats = self._out_stats, None
        return y


class EncBlock(tnn.Module):
    """stride-2 4x4 convolution, BatchNorm, GELU (`down`), then `num_blocks` bottlenecks (`res`) — reference :27-36"""

    def __init__(self, in_ch: int, out_ch: int, num_blocks: int = 2, groups: int = 8):
        super().__init__()
        halve = wnn.Conv2d(in_ch, out_ch, 4, stride=2, padding=1, bias=False)
        self.down = tnn.Sequential(halve, wnn.BatchNorm2d(out_ch), wnn.GELU())
        self.res = _residual_stack(out_ch, num_blocks, groups)

    def forward(self, x):
        conv, bn, _ = self.down
        return self.res(_with_unit_stats(Fn.DownUnitFn.apply(x, conv.weight, bn.weight, bn.bias, bn), bn))


class DecBlock(tnn.Module):
    """stride-2 4x4 transposed convolution, BatchNorm, GELU (`up`), then the bottlenecks (`res`) — reference :38-47"""

    def __init__(self, in_ch: int, out_ch: int, num_blocks: int = 2, groups: int = 8):
        super().__init__()
        double = wnn.ConvTranspose2d(in_ch, out_ch, 4, stride=2, padding=1, bias=False)
        self.up = tnn.Sequential(double, wnn.BatchNorm2d(out_ch), wnn.GELU())
        self.res = _residual_stack(out_ch, num_blocks, groups)

    def forward(self, x):
        convt, bn, _ = self.up
        return self.res(_with_unit_stats(Fn.UpUnitFn.apply(x, convt.weight, bn.weight, bn.bias, bn), bn))


class PosAwareAE_TF(tnn.Module):
    """1 x H x W frame -> `latent_dim` vector -> 1 x H x W reconstruction in (0, 1) — reference :52-106.
    `enc`: four EncBlocks (1/16 resolution, 1024 channels) + 1x1 to `latent_channels`, learned `pos_emb` added,
    `to_latent` / `from_latent` linear bottleneck, `dec`: 1x1, four DecBlocks, 3x3 to the image, sigmoid."""

    def __init__(self, in_channels: int = 1, latent_channels: int = 64, groups: int = 8,
                 latent_dim: int = 2048, *, img_size: int = 128):
        super().__init__()
        if img_size % 16:
            raise ValueError("img_size must be a multiple of 16 (four stride-2 stages)")
        self.img_size, self.latent_channels = img_size, latent_channels
        side = self.latent_hw = img_size // 16
        flat = side * side * latent_channels

        stages, c_prev = [], in_channels
        for width in _ENC_WIDTHS:
            stages.append(EncBlock(c_prev, width, num_blocks=_BLOCKS_PER_STAGE, groups=groups))
            c_prev = width
        self.enc = tnn.Sequential(*stages, wnn.Conv2d(c_prev, latent_channels, 1))
        self.pos_emb = tnn.Parameter(torch.randn(1, latent_channels, side, side))
        self.to_latent = wnn.Linear(flat, latent_dim)
        self.from_latent = wnn.Linear(latent_dim, flat)

        stages, c_prev = [wnn.Conv2d(latent_channels, _DEC_WIDTHS[0], 1)], _DEC_WIDTHS[0]
        for width in _DEC_WIDTHS:
            stages.append(DecBlock(c_prev, width, num_blocks=_BLOCKS_PER_STAGE, groups=groups))
            c_prev = width
        self.dec = tnn.Sequential(*stages, wnn.Conv2d(c_prev, in_channels, 3, padding=1))
        self.act = wnn.Sigmoid()

    def encode(self, x):
        """frames -> latent vector (reference :88-94); the broadcast `pos_emb` add rides in the epilogue of the
        1x1 convolution that closes `enc`"""
        *blocks, to_channels = self.enc
        for blk in blocks:
            x = blk(x)
        grid = Fn.Conv1x1Fn.apply(x, to_channels.weight, to_channels.bias, self.pos_emb)
        return self.to_latent(grid.flatten(1))

    def decode(self, z_flat):
        """latent vector -> reconstruction (reference :96-102)"""
        grid = self.from_latent(z_flat).view(z_flat.size(0), self.latent_channels, self.latent_hw, self.latent_hw)
        return self.act(self.dec(grid))

    def forward(self, x):
        z = self.encode(x)
        return self.decode(z), z
