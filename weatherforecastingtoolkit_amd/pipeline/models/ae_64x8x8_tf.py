"""Drop-in for the reference's pipeline/models/ae_64x8x8_tf.py — the file that
experiments/ae_v2/train.py:18 actually imports: the conv autoencoder of
ae_64x8x8_lin plus an 8-layer nn.TransformerEncoder on the 64 latent tokens in
`decode` (reference :77-80, :101-112).  743 state_dict entries incl. the unused template
layer `tf_encoder.*`.

Reference quirk kept on purpose: the encoder layers are seq-first but are fed
(B, 64, C), so attention runs ACROSS THE BATCH (sequence length = B) and the 64
tokens act as the batch.  Results therefore depend on how a
global batch is sharded across GPUs, exactly as with the reference under DDP.
`tf_encoder.*` never receives gradients (it is only the template that
nn.TransformerEncoder deep-copies).
"""
from __future__ import annotations

import torch
import torch.nn as tnn

from ... import functional as Fn
from ... import nn as wnn
from .ae_64x8x8_lin import Bottleneck, DecBlock, EncBlock  # noqa: F401  (same building blocks)


class PosAwareAE_TF(tnn.Module):
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

        self.tf_encoder = wnn.TransformerEncoderLayer(d_model=latent_channels, nhead=8, dim_feedforward=2048,
                                                      dropout=0.1)
        self.tf = wnn.TransformerEncoder(self.tf_encoder, num_layers=8, enable_nested_tensor=False)

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
        z_tokens = z.flatten(2).transpose(1, 2)      # (B, hw*hw, C) — layout change only
        z_tokens = self.tf(z_tokens)                 # seq-first layers: attention across B
        z = z_tokens.transpose(1, 2).reshape(B, self.latent_channels, self.latent_hw, self.latent_hw)
        return self.act(self.dec(z))

    def forward(self, x):
        z = self.encode(x)
        return self.decode(z), z
