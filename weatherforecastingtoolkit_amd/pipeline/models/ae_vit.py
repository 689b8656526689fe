"""Path-B token autoencoder on libwfae.so kernels.

Mirrors `pipeline/models/ae_vit.py` of the reference: `GlobalCrossEncode` (:4-42), `GlobalCrossDecode` (:44-82)
and `AE_ViT_2048` (:84-162) with the same constructor signatures, sub-module / parameter names and state_dict
keys.  16x16 patch embedding and un-patching are patch (un)folding + one MFMA GEMM each; the twelve
TransformerEncoderLayer(d=512, 8 heads, ff=2048, GELU, batch_first=True) blocks run on the layernorm / attention /
linear kernels; GlobalCrossEncode is a single-query attention kernel.

GlobalCrossDecode attends L token queries to ONE key/value (the latent): softmax over a single key is 1, so
the output is `out(v)` for every token and the query path (q_proj, dec_queries, the k half of kv_proj) gets an
exactly-zero gradient in the reference; here that is what is computed — v -> out -> broadcast over the tokens,
zero gradients written for the dead parameters.
"""
from __future__ import annotations

import torch
import torch.nn as tnn

from ... import functional as Fn
from ... import nn as wnn
from ..._lib import WfaeError


class GlobalCrossEncode(tnn.Module):
    """q: (B, 1, d_latent), kv: (B, L, d_token) -> (B, d_latent)"""

    def __init__(self, d_token, d_latent, n_heads=8):
        super().__init__()
        assert d_latent % n_heads == 0 and d_token % n_heads == 0
        self.nh = n_heads
        self.dh_q = d_latent // n_heads
        self.dh_kv = d_token // n_heads
        self.scale = self.dh_q ** -0.5
        self.q_proj = wnn.Linear(d_latent, d_latent)
        self.kv_proj = wnn.Linear(d_token, 2 * d_latent)
        self.out = wnn.Linear(d_latent, d_latent)

    def forward(self, q, kv):
        b, l, _ = kv.shape
        if q.shape[1] != 1:
            raise WfaeError("GlobalCrossEncode: one query per batch element")
        qp = self.q_proj(q.reshape(b, -1))                       # (B, d_latent)
        kvp = self.kv_proj(kv.reshape(b * l, -1))                # (B*L, 2*d_latent) = [k | v]
        o = Fn.SingleQueryAttnFn.apply(qp, kvp, l, self.nh)      # (B, d_latent)
        return self.out(o)


class _DeadGradFn(torch.autograd.Function):
    """identity on x that also hands exact-zero gradients to parameters the reference's graph reaches only
    through softmax over a single key"""

    @staticmethod
    def forward(ctx, x, *dead):
        ctx.dead = dead
        return x.view_as(x)

    @staticmethod
    def backward(ctx, dy):
        zeros = []
        for p in ctx.dead:
            z = Fn.grad_buffer(p)
            z.zero_()
            zeros.append(z)
        return (dy, *zeros)


class GlobalCrossDecode(tnn.Module):
    """q: (B, L, d_token), kv: (B, 1, d_latent) -> (B, L, d_token)"""

    def __init__(self, d_token, d_latent, n_heads=8):
        super().__init__()
        assert d_latent % n_heads == 0 and d_token % n_heads == 0
        self.nh = n_heads
        self.dh_q = d_token // n_heads
        self.dh_kv = d_latent // n_heads
        self.scale = self.dh_kv ** -0.5
        self.q_proj = wnn.Linear(d_token, d_token)
        self.kv_proj = wnn.Linear(d_latent, 2 * d_token)
        self.out = wnn.Linear(d_token, d_token)

    def forward(self, q, kv, _dead_params=()):
        b, l, e = q.shape
        if kv.shape[1] != 1:
            raise WfaeError("GlobalCrossDecode: one key/value per batch element")
        kvp = self.kv_proj(kv.reshape(b, -1))                    # (B, 2*d_token) = [k | v]
        v = Fn.SliceColsFn.apply(kvp, e, 2 * e)                  # attention weights are all 1: out = v
        o = self.out(v)                                          # (B, d_token), the same for every token
        o = _DeadGradFn.apply(o, self.q_proj.weight, self.q_proj.bias, *_dead_params)
        return Fn.RepeatRowsFn.apply(o, l).view(b, l, e)


# geometry of AE_ViT_2048 (reference :88-94): 128x128 frames, 16x16 patches -> 8x8 = 64 tokens of width 512,
# one 2048-wide latent, six transformer blocks on each side, eight heads
_IMG, _PATCH, _CH = 128, 16, 1
_D_TOKEN, _D_LATENT = 512, 2048
_DEPTH_ENC = _DEPTH_DEC = 6
_HEADS = 8


def _transformer(depth):
    """`depth` post-norm GELU blocks (d_model 512, ff 2048, dropout 0.1, batch_first) — reference :104-109, :118-123"""
    block = wnn.TransformerEncoderLayer(d_model=_D_TOKEN, nhead=_HEADS, dim_feedforward=4 * _D_TOKEN, dropout=0.1,
                                        activation="gelu", batch_first=True)
    return wnn.TransformerEncoder(block, depth, enable_nested_tensor=False)


class AE_ViT_2048(tnn.Module):
    def __init__(self):
        super().__init__()
        self.seq = _IMG // _PATCH
        self.d_token, self.d_latent = _D_TOKEN, _D_LATENT
        n_tok = self.seq ** 2
        # registration order = the reference's state_dict order
        self.patch_embed = tnn.Conv2d(_CH, _D_TOKEN, _PATCH, _PATCH)            # parameters only; forward below
        self.pos_embed = tnn.Parameter(torch.randn(1, n_tok, _D_TOKEN))
        self.encoder = _transformer(_DEPTH_ENC)
        self.query_vec = tnn.Parameter(torch.randn(1, 1, _D_LATENT))
        self.to_latent = GlobalCrossEncode(_D_TOKEN, _D_LATENT, n_heads=_HEADS)
        self.dec_queries = tnn.Parameter(torch.randn(1, n_tok, _D_TOKEN))
        self.from_latent = GlobalCrossDecode(_D_TOKEN, _D_LATENT, n_heads=_HEADS)
        self.decoder = _transformer(_DEPTH_DEC)
        self.unpatch = tnn.ConvTranspose2d(_D_TOKEN, _CH, _PATCH, _PATCH)       # parameters only

    def encode_tokens(self, x):
        b = x.size(0)
        n = self.seq * self.seq
        z = Fn.PatchEmbedFn.apply(x, self.patch_embed.weight, self.patch_embed.bias)   # (B*64, 512)
        z = Fn.AddBcastFn.apply(z.view(b, n, self.d_token), self.pos_embed)
        return self.encoder(z)

    def forward(self, x):
        b = x.size(0)
        n = self.seq * self.seq
        z = self.encode_tokens(x)
        q = Fn.ExpandRowsFn.apply(self.query_vec, b).view(b, 1, self.d_latent)
        latent = self.to_latent(q, z)                                                  # (B, 2048)
        dec_q = self.dec_queries.expand(b, -1, -1)               # shapes only: its gradient is exactly zero
        z_dec = self.from_latent(dec_q, latent.unsqueeze(1), _dead_params=(self.dec_queries,))
        z_dec = Fn.AddBcastFn.apply(z_dec, self.pos_embed)
        z_dec = self.decoder(z_dec)
        out = Fn.UnpatchFn.apply(z_dec.reshape(b * n, self.d_token), self.unpatch.weight, self.unpatch.bias,
                                 b, self.seq, self.seq)
        return out, latent
