"""Layer classes with the reference's constructor signatures and state_dict
keys (they subclass the torch.nn containers for parameter registration and
default initialisation only — SURVEY.md Appendix B.9); every forward/backward
runs on libwfae.so kernels through functional.py.
"""
from __future__ import annotations

import torch
import torch.nn as tnn

from . import functional as Fn
from ._lib import WfaeError


def _pair(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (v, v)


class Conv2d(tnn.Conv2d):
    """nn.Conv2d restricted to the geometries the hot path uses:
    1x1 (ae_64x8x8_lin.py:15,19,69,79), 4x4 s2 p1 (:31), 3x3 s1 p1 any groups (:17,84), and the PatchGAN
    discriminator's 4x4 s2/s1 p1 (+bias) and zero-padded 1x1 (autoencoderkl/losses/model.py:125-145)."""

    def forward(self, x):
        k, s, p = _pair(self.kernel_size), _pair(self.stride), _pair(self.padding)
        if k == (1, 1) and s == (1, 1) and p == (0, 0) and self.groups == 1:
            return Fn.Conv1x1Fn.apply(x, self.weight, self.bias, None)
        if k == (3, 3) and s == (1, 1) and p == (1, 1):
            return Fn.DConvFn.apply(x, self.weight, self.bias, self.groups)
        if k == (4, 4) and s == (2, 2) and p == (1, 1) and self.groups == 1 and self.bias is None:
            return Fn.Conv4x4DownFn.apply(x, self.weight)
        if k == (4, 4) and s in ((1, 1), (2, 2)) and p == (1, 1) and self.groups == 1:
            return Fn.Conv4Fn.apply(x, self.weight, self.bias, s[0])
        if k == (1, 1) and s == (1, 1) and p[0] == p[1] and self.groups == 1 and self.padding_mode == "zeros":
            return Fn.Conv1x1Fn.apply(Fn.Pad2dFn.apply(x, p[0]), self.weight, self.bias, None)
        raise WfaeError(f"Conv2d geometry k={k} s={s} p={p} g={self.groups} bias={self.bias is not None} "
                        "has no gfx950 kernel in this build")


class ConvTranspose2d(tnn.ConvTranspose2d):
    """nn.ConvTranspose2d(Cin, Cout, 4, stride=2, padding=1, bias=False) (ae_64x8x8_lin.py:42)."""

    def forward(self, x, output_size=None):
        k, s, p = _pair(self.kernel_size), _pair(self.stride), _pair(self.padding)
        if k == (4, 4) and s == (2, 2) and p == (1, 1) and self.groups == 1 and self.bias is None \
                and _pair(self.output_padding) == (0, 0):
            return Fn.ConvT4x4UpFn.apply(x, self.weight)
        raise WfaeError("ConvTranspose2d geometry has no gfx950 kernel in this build")


class BatchNorm2d(tnn.BatchNorm2d):
    """nn.BatchNorm2d(eps=1e-5, momentum=0.1).  num_batches_tracked is counted on
    the host and materialised into the buffer when the state_dict is read."""

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self._nbt_pending = 0
        if self.momentum is None or not self.affine or not self.track_running_stats:
            raise WfaeError("BatchNorm2d: only affine, running-stat, fixed-momentum BN is built")

    def flush(self):
        if self._nbt_pending:
            self.num_batches_tracked += self._nbt_pending
            self._nbt_pending = 0

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        self.flush()
        super()._save_to_state_dict(destination, prefix, keep_vars)

    def _load_from_state_dict(self, *a, **kw):
        self._nbt_pending = 0
        super()._load_from_state_dict(*a, **kw)

    def forward(self, x, act=0):
        return Fn.BatchNormActFn.apply(x, self.weight, self.bias, self, act)


class GELU(tnn.GELU):
    def forward(self, x):
        if self.approximate != "none":
            raise WfaeError("only exact (erf) GELU is built")
        return Fn.GeluFn.apply(x)


class LeakyReLU(tnn.LeakyReLU):
    """nn.LeakyReLU(0.2[, inplace]) — the only slope the reference uses (autoencoderkl/losses/model.py:125-141)."""

    def __init__(self, negative_slope=0.2, inplace=False):
        super().__init__(negative_slope, inplace)
        if abs(negative_slope - 0.2) > 1e-12:
            raise WfaeError("LeakyReLU: only negative_slope=0.2 is built")

    def forward(self, x):
        return Fn.LeakyReluFn.apply(x)


class Sigmoid(tnn.Sigmoid):
    def forward(self, x):
        return Fn.SigmoidFn.apply(x)


class Linear(tnn.Linear):
    def forward(self, x):
        shp = x.shape
        y = Fn.LinearFn.apply(x.reshape(-1, shp[-1]), self.weight, self.bias)
        return y.view(*shp[:-1], self.out_features)


class TransformerEncoderLayer(tnn.TransformerEncoderLayer):
    """nn.TransformerEncoderLayer(d_model, nhead, dim_feedforward, dropout[, activation, batch_first]) as the
    reference builds it: post-norm, ReLU, batch_first=False in pipeline/models/ae_64x8x8_tf.py:77-79; post-norm,
    GELU, batch_first=True in pipeline/models/ae_vit.py:104-109,118-123.  Parameters and state_dict keys are
    torch's; forward/backward run on libwfae.so kernels."""

    def forward(self, src, src_mask=None, src_key_padding_mask=None, is_causal=False):
        if src_mask is not None or src_key_padding_mask is not None or is_causal:
            raise WfaeError("TransformerEncoderLayer: masks are not built (the reference passes none)")
        at = self.self_attn
        if self.norm_first or not at._qkv_same_embed_dim or at.bias_k is not None:
            raise WfaeError("TransformerEncoderLayer: only the post-norm configuration is built")
        act = getattr(self, "activation_relu_or_gelu", 1)
        if act not in (1, 2):
            raise WfaeError("TransformerEncoderLayer: only ReLU and GELU are built")
        bf = bool(at.batch_first)
        if bf:
            n, s, e = src.shape      # (batch, tokens, E): attention over the tokens of each batch element
        else:
            s, n, e = src.shape      # (seq, batch, E)
        h = at.num_heads
        tr = self.training
        x = src.contiguous().view(s * n, e)
        qkv = Fn.LinearFn.apply(x, at.in_proj_weight, at.in_proj_bias)
        p_attn = float(at.dropout) if tr else 0.0
        a = Fn.MhaSeqFirstFn.apply(qkv, s, n, h, p_attn, Fn.next_seed() if p_attn > 0 else 0, bf)
        a = Fn.LinearFn.apply(a, at.out_proj.weight, at.out_proj.bias)
        a = Fn.dropout(a, self.dropout1.p, tr)
        x1 = Fn.AddLayerNormFn.apply(x, a, self.norm1.weight, self.norm1.bias, self.norm1.eps)
        f = Fn.LinearFn.apply(x1, self.linear1.weight, self.linear1.bias)
        f = Fn.dropout(Fn.ReluFn.apply(f) if act == 1 else Fn.GeluFn.apply(f), self.dropout.p, tr)
        f = Fn.LinearFn.apply(f, self.linear2.weight, self.linear2.bias)
        f = Fn.dropout(f, self.dropout2.p, tr)
        x2 = Fn.AddLayerNormFn.apply(x1, f, self.norm2.weight, self.norm2.bias, self.norm2.eps)
        return x2.view(n, s, e) if bf else x2.view(s, n, e)


class TransformerEncoder(tnn.TransformerEncoder):
    """nn.TransformerEncoder(layer, num_layers) — container only; skips torch's nested-tensor fast path."""

    def forward(self, src, mask=None, src_key_padding_mask=None, is_causal=None):
        out = src
        for mod in self.layers:
            out = mod(out)
        if self.norm is not None:
            out = self.norm(out)
        return out


def flush_bn_counters(module: torch.nn.Module):
    for m in module.modules():
        if isinstance(m, BatchNorm2d):
            m.flush()
