"""Counter-based synthetic data: weights and SEVIR-shaped frames.

Everything here is a pure function of (seed, key, index), implemented with
numpy uint64 arithmetic (splitmix64), so the same tensors can be rebuilt in
the build container (where the golden vectors are made from the reference)
and on the GPU box (where the reference does not exist) without relying on
any torch RNG stream.

Distributions follow the torch defaults the reference relies on
(SURVEY.md Appendix B.9): conv / linear weights and biases are
U(-1/sqrt(fan_in), 1/sqrt(fan_in)); BatchNorm gamma/beta are perturbed away
from (1, 0) so parity tests exercise the affine part; pos_emb is ~N(0,1)
(reference: pipeline/models/ae_64x8x8_lin.py:72).
"""
from __future__ import annotations

import zlib
from collections import OrderedDict

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def _stream(seed: int, key: str) -> np.uint64:
    h = zlib.crc32(key.encode("utf-8")) & 0xFFFFFFFF
    base = np.array([(seed & 0xFFFFFFFF) << 32 | h], dtype=np.uint64)
    return _splitmix64(base)[0]


def uniform01(seed: int, key: str, n: int) -> np.ndarray:
    """n floats in [0,1) with 24 random bits each (exactly representable)."""
    with np.errstate(over="ignore"):
        s = _stream(seed, key)
        idx = np.arange(n, dtype=np.uint64) * np.uint64(0xD1342543DE82EF95) + s
        z = _splitmix64(idx)
    return ((z >> np.uint64(40)).astype(np.float32)) * np.float32(1.0 / (1 << 24))


def uniform(seed: int, key: str, shape, lo: float, hi: float) -> np.ndarray:
    n = int(np.prod(shape)) if len(shape) else 1
    u = uniform01(seed, key, n)
    return (np.float32(lo) + (np.float32(hi) - np.float32(lo)) * u).reshape(shape).astype(np.float32)


def normal(seed: int, key: str, shape) -> np.ndarray:
    n = int(np.prod(shape)) if len(shape) else 1
    u1 = uniform01(seed, key + "/a", n).astype(np.float64)
    u2 = uniform01(seed, key + "/b", n).astype(np.float64)
    r = np.sqrt(-2.0 * np.log(1.0 - u1))
    return (r * np.cos(2.0 * np.pi * u2)).reshape(shape).astype(np.float32)


# --------------------------------------------------------------------------
# state_dict layout of PosAwareAE_TF (reference
# pipeline/models/ae_64x8x8_lin.py:52-86).  Pure shape arithmetic.
# --------------------------------------------------------------------------
ENC_CH = (256, 512, 1024, 1024)
DEC_CH = (1024, 512, 256, 128)


def _bn_entries(prefix: str, c: int):
    return [
        (prefix + ".weight", (c,), "bn_w"),
        (prefix + ".bias", (c,), "bn_b"),
        (prefix + ".running_mean", (c,), "bn_rm"),
        (prefix + ".running_var", (c,), "bn_rv"),
        (prefix + ".num_batches_tracked", (), "bn_n"),
    ]


def _bottleneck_entries(prefix: str, c: int, groups: int):
    mid = c // 4
    g = min(groups, mid)
    e = []
    e += _bn_entries(prefix + ".f.0", c)
    e.append((prefix + ".f.2.weight", (mid, c, 1, 1), "conv"))
    e += _bn_entries(prefix + ".f.3", mid)
    e.append((prefix + ".f.5.weight", (mid, mid // g, 3, 3), "conv"))
    e += _bn_entries(prefix + ".f.6", mid)
    e.append((prefix + ".f.8.weight", (c, mid, 1, 1), "conv"))
    return e


def ae_state_dict_spec(img_size: int = 128, in_channels: int = 1, latent_channels: int = 64,
                       groups: int = 8, latent_dim: int = 2048, num_blocks: int = 4,
                       enc_ch=ENC_CH, dec_ch=DEC_CH):
    """[(key, shape, kind)] in the registration order of the reference module."""
    hw = img_size // 16
    e = []
    cin = in_channels
    for i, c in enumerate(enc_ch):
        e.append((f"enc.{i}.down.0.weight", (c, cin, 4, 4), "conv"))
        e += _bn_entries(f"enc.{i}.down.1", c)
        for j in range(num_blocks):
            e += _bottleneck_entries(f"enc.{i}.res.{j}", c, groups)
        cin = c
    n_enc = len(enc_ch)
    e.append((f"enc.{n_enc}.weight", (latent_channels, cin, 1, 1), "conv"))
    e.append((f"enc.{n_enc}.bias", (latent_channels,), ("bias", cin)))
    spec = [("pos_emb", (1, latent_channels, hw, hw), "normal")] + e
    feat = hw * hw * latent_channels
    spec.append(("to_latent.weight", (latent_dim, feat), "linear"))
    spec.append(("to_latent.bias", (latent_dim,), ("bias", feat)))
    spec.append(("from_latent.weight", (feat, latent_dim), "linear"))
    spec.append(("from_latent.bias", (feat,), ("bias", latent_dim)))
    d = []
    d.append(("dec.0.weight", (dec_ch[0], latent_channels, 1, 1), "conv"))
    d.append(("dec.0.bias", (dec_ch[0],), ("bias", latent_channels)))
    cin = dec_ch[0]
    for i, c in enumerate(dec_ch):
        k = i + 1
        # ConvTranspose2d weight is (Cin, Cout, kH, kW); torch computes its
        # fan_in from dim 1 (= Cout) * receptive field.
        d.append((f"dec.{k}.up.0.weight", (cin, c, 4, 4), "conv"))
        d += _bn_entries(f"dec.{k}.up.1", c)
        for j in range(num_blocks):
            d += _bottleneck_entries(f"dec.{k}.res.{j}", c, groups)
        cin = c
    k = len(dec_ch) + 1
    d.append((f"dec.{k}.weight", (in_channels, cin, 3, 3), "conv"))
    d.append((f"dec.{k}.bias", (in_channels,), ("bias", cin * 9)))
    return spec + d


def _tf_layer_entries(prefix: str, d: int = 64, ff: int = 2048):
    return [
        (prefix + ".self_attn.in_proj_weight", (3 * d, d), "linear"),
        (prefix + ".self_attn.in_proj_bias", (3 * d,), ("bias", d)),
        (prefix + ".self_attn.out_proj.weight", (d, d), "linear"),
        (prefix + ".self_attn.out_proj.bias", (d,), ("bias", d)),
        (prefix + ".linear1.weight", (ff, d), "linear"),
        (prefix + ".linear1.bias", (ff,), ("bias", d)),
        (prefix + ".linear2.weight", (d, ff), "linear"),
        (prefix + ".linear2.bias", (d,), ("bias", ff)),
        (prefix + ".norm1.weight", (d,), "bn_w"),
        (prefix + ".norm1.bias", (d,), "bn_b"),
        (prefix + ".norm2.weight", (d,), "bn_w"),
        (prefix + ".norm2.bias", (d,), "bn_b"),
    ]


def ae_tf_state_dict_spec(img_size: int = 128, num_layers: int = 8):
    """state_dict of pipeline/models/ae_64x8x8_tf.py::PosAwareAE_TF (743 entries): the `_lin` layout
    with `tf_encoder.*` and `tf.layers.{i}.*` registered between from_latent and dec."""
    base = ae_state_dict_spec(img_size)
    cut = next(i for i, (k, _, _) in enumerate(base) if k == "dec.0.weight")
    tf = _tf_layer_entries("tf_encoder")
    for i in range(num_layers):
        tf += _tf_layer_entries(f"tf.layers.{i}")
    return base[:cut] + tf + base[cut:]


def disc_state_dict_spec(input_nc: int = 1, ndf: int = 64, n_layers: int = 3):
    """state_dict of NLayerDiscriminator (reference pipeline/models/autoencoderkl/losses/model.py:100-150)
    after `.apply(weights_init)` (:6-12): conv weights ~N(0, 0.02), BN gamma ~N(1, 0.02), beta 0,
    conv biases at the torch default."""
    e = [("main.0.weight", (ndf, input_nc, 4, 4), "gan_conv"), ("main.0.bias", (ndf,), ("bias", input_nc * 16))]
    idx, mult = 2, 1
    for n in range(1, n_layers + 1):
        prev, mult = mult, min(2 ** n, 8)
        e.append((f"main.{idx}.weight", (ndf * mult, ndf * prev, 4, 4), "gan_conv"))
        c = ndf * mult
        e += [(f"main.{idx + 1}.weight", (c,), "gan_bn_w"), (f"main.{idx + 1}.bias", (c,), "zeros"),
              (f"main.{idx + 1}.running_mean", (c,), "zeros"), (f"main.{idx + 1}.running_var", (c,), "ones"),
              (f"main.{idx + 1}.num_batches_tracked", (), "bn_n")]
        idx += 3
    e += [(f"main.{idx}.weight", (1, ndf * mult, 1, 1), "gan_conv"), (f"main.{idx}.bias", (1,), ("bias", ndf * mult))]
    return e


def generic_state_dict(named_shapes, seed: int = 0) -> "OrderedDict[str, np.ndarray]":
    """Synthetic values for an arbitrary state_dict layout [(key, shape)] (used for AE_ViT_2048, whose 163 keys
    are read off the module itself): matrices / conv kernels U(+-1/sqrt(fan_in)), LayerNorm weights U(0.8,1.2),
    other vectors U(-0.1,0.1), 3-D parameters (positional / query tokens) ~N(0,1)."""
    out = OrderedDict()
    for key, shape in named_shapes:
        shape = tuple(shape)
        if len(shape) == 3:
            out[key] = normal(seed, key, shape)
        elif len(shape) in (2, 4):
            b = 1.0 / np.sqrt(int(np.prod(shape[1:])))
            out[key] = uniform(seed, key, shape, -b, b)
        elif len(shape) == 1 and ".norm" in key and key.endswith(".weight"):
            out[key] = uniform(seed, key, shape, 0.8, 1.2)
        elif len(shape) == 1:
            out[key] = uniform(seed, key, shape, -0.1, 0.1)
        else:
            raise ValueError(f"{key}: shape {shape}")
    return out


def synth_tensor(seed: int, key: str, shape, kind) -> np.ndarray:
    if kind == "gan_conv":
        return (np.float32(0.02) * normal(seed, key, shape)).astype(np.float32)
    if kind == "gan_bn_w":
        return (np.float32(1.0) + np.float32(0.02) * normal(seed, key, shape)).astype(np.float32)
    if kind == "zeros":
        return np.zeros(shape, dtype=np.float32)
    if kind == "ones":
        return np.ones(shape, dtype=np.float32)
    if isinstance(kind, tuple) and kind[0] == "bias":
        b = 1.0 / np.sqrt(kind[1])
        return uniform(seed, key, shape, -b, b)
    if kind == "conv":
        fan_in = int(np.prod(shape[1:]))
        b = 1.0 / np.sqrt(fan_in)
        return uniform(seed, key, shape, -b, b)
    if kind == "linear":
        b = 1.0 / np.sqrt(shape[1])
        return uniform(seed, key, shape, -b, b)
    if kind == "normal":
        return normal(seed, key, shape)
    if kind == "bn_w":
        return uniform(seed, key, shape, 0.8, 1.2)
    if kind == "bn_b":
        return uniform(seed, key, shape, -0.1, 0.1)
    if kind == "bn_rm":
        return uniform(seed, key, shape, -0.05, 0.05)
    if kind == "bn_rv":
        return uniform(seed, key, shape, 0.9, 1.1)
    if kind == "bn_n":
        return np.zeros((), dtype=np.int64)
    raise ValueError(kind)


def synth_state_dict(spec, seed: int = 0) -> "OrderedDict[str, np.ndarray]":
    return OrderedDict((k, synth_tensor(seed, k, shp, kind)) for k, shp, kind in spec)


# --------------------------------------------------------------------------
# SEVIR-shaped frames.
# --------------------------------------------------------------------------
def blob_events(n_events: int, size: int, frames: int, seed: int = 1234) -> np.ndarray:
    """uint8 VIL-like event array (n_events, H, W, T): per frame 6 Gaussian
    blobs, clipped to 255 and floored, roughly half zeros like real VIL.
    Layout matches the HDF5 'vil' dataset the reference loader reads
    (pipeline/datasets/sevire/sevir.py:453-482)."""
    yy, xx = np.meshgrid(np.arange(size, dtype=np.float32), np.arange(size, dtype=np.float32), indexing="ij")
    out = np.zeros((n_events, size, size, frames), dtype=np.uint8)
    for e in range(n_events):
        for t in range(frames):
            key = f"blob/{e}/{t}"
            p = uniform01(seed, key, 24)
            f = np.zeros((size, size), dtype=np.float32)
            for b in range(6):
                cy, cx = p[4 * b] * size, p[4 * b + 1] * size
                sg = size / 32 + p[4 * b + 2] * (size / 8 - size / 32)
                amp = 64 + p[4 * b + 3] * (255 - 64)
                f += amp * np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * sg * sg))
            f = np.clip(f - 24.0, 0, 255)
            out[e, :, :, t] = np.floor(f).astype(np.uint8)
    return out


def uniform_frames(batch: int, size: int, seed: int = 1234, key: str = "frames") -> np.ndarray:
    """(B,1,H,W) fp32 frames u8/255 with u8 ~ U{0..255}: the throughput input."""
    u = uniform01(seed, key, batch * size * size)
    return (np.floor(u * 256.0).astype(np.float32) / np.float32(255.0)).reshape(batch, 1, size, size)
