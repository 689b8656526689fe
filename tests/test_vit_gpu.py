"""GPU parity of the Path-B token autoencoder AE_ViT_2048 (reference pipeline/models/ae_vit.py:84-162) against
tests/golden/g10_vit128_b3.npz, produced from the REAL reference module (eval forward; train forward/backward with
every dropout probability set to 0)."""
import numpy as np
import pytest
import torch

from tests._util import golden, relerr

pytestmark = pytest.mark.gpu


def _build(dev):
    from weatherforecastingtoolkit_amd import synth
    from weatherforecastingtoolkit_amd.pipeline.models.ae_vit import AE_ViT_2048
    g = golden("g10_vit128_b3")
    net = AE_ViT_2048()
    keys = [str(k) for k in g["keys"]]
    assert list(net.state_dict().keys()) == keys                       # the reference's 163 keys, same order
    shapes = [(k, tuple(v.shape)) for k, v in net.state_dict().items()]
    np_sd = synth.generic_state_dict(shapes, seed=3)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in np_sd.items()}, strict=True)
    return net.to(dev), g


def test_vit_eval_forward_golden(dev):
    from weatherforecastingtoolkit_amd import synth
    net, g = _build(dev)
    net.eval()
    x = torch.from_numpy(synth.uniform_frames(3, 128, seed=1234)).to(dev)
    with torch.no_grad():
        out, z = net(x)
    idx = torch.from_numpy(g["lattice"]).to(dev)
    assert tuple(out.shape) == (3, 1, 128, 128) and tuple(z.shape) == (3, 2048)
    assert relerr(z, g["eval_latent"]) < 1e-4
    assert relerr(out[:, 0][:, idx][:, :, idx], g["eval_out_lattice"]) < 1e-4


def test_vit_train_step_golden(dev):
    from weatherforecastingtoolkit_amd import functional as Fn
    from weatherforecastingtoolkit_amd import synth
    net, g = _build(dev)
    net.train()
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if isinstance(m, torch.nn.MultiheadAttention):
            m.dropout = 0.0
    x = torch.from_numpy(synth.uniform_frames(3, 128, seed=1234)).to(dev)
    out, z = net(x)
    loss = Fn.mse_loss(out, x)
    loss.backward()
    idx = torch.from_numpy(g["lattice"]).to(dev)
    assert relerr(z, g["latent"]) < 1e-4
    assert relerr(out[:, 0][:, idx][:, :, idx], g["out_lattice"]) < 1e-4
    assert abs(loss.item() - float(g["loss"])) < 1e-5 * float(g["loss"])
    names = [str(n) for n in g["grad_names"]]
    params = dict(net.named_parameters())
    assert list(params) == names
    worst = 0.0
    for n, ref in zip(names, g["grad_norms"]):
        p = params[n]
        assert p.grad is not None, n
        got = p.grad.double().norm().item()
        if ref == 0.0:
            assert got == 0.0, n          # dec_queries / from_latent.q_proj: softmax over ONE key has zero gradient
        else:
            worst = max(worst, abs(got - ref) / ref)
            assert abs(got - ref) < 1e-3 * ref, (n, got, ref)
    for k in g.files:
        if k.startswith("grad/"):
            assert relerr(params[k[5:]].grad, g[k]) < 1e-3, k
        elif k.startswith("grad_head/"):
            assert relerr(params[k[10:]].grad.flatten()[:4096], g[k]) < 1e-3, k


def test_vit_pieces_vs_torch(dev):
    """single-query attention, batch-first attention with head dim 64, patch (un)folding against torch"""
    import torch.nn.functional as F
    from weatherforecastingtoolkit_amd import ops
    torch.manual_seed(0)
    # batch-first MHA, 8 heads x 64
    b, l, h, d = 3, 64, 8, 64
    qkv = torch.randn(b * l, 3 * h * d) * 0.5
    dout = torch.randn(b * l, h * d)
    qr = qkv.clone().requires_grad_(True)
    q, k, v = [t.view(b, l, h, d).transpose(1, 2) for t in qr.split(h * d, dim=1)]
    ref = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(b * l, h * d)
    ref.backward(dout)
    out, probs = ops.mha_fwd(qkv.to(dev), l, b, h, d, 0.0, 0, True)
    assert relerr(out, ref) < 2e-5
    assert relerr(ops.mha_bwd(qkv.to(dev), probs, dout.to(dev), l, b, h, d, 0.0, 0, True), qr.grad) < 1e-4
    # single-query attention, 8 heads x 256
    b, l, h, d = 4, 64, 8, 256
    qq, kv, do = torch.randn(b, h * d) * 0.3, torch.randn(b * l, 2 * h * d) * 0.3, torch.randn(b, h * d)
    qg, kvg = qq.clone().requires_grad_(True), kv.clone().requires_grad_(True)
    kk = kvg.view(b, l, 2, h, d)
    att = (qg.view(b, 1, h, d).transpose(1, 2) @ kk[:, :, 0].transpose(1, 2).transpose(-2, -1)) * d ** -0.5
    ref = (att.softmax(-1) @ kk[:, :, 1].transpose(1, 2)).transpose(1, 2).reshape(b, h * d)
    ref.backward(do)
    out, probs = ops.sq_attn_fwd(qq.to(dev), kv.to(dev), b, l, h, d)
    assert relerr(out, ref) < 2e-5
    dq, dkv = ops.sq_attn_bwd(qq.to(dev), kv.to(dev), probs, do.to(dev), b, l, h, d)
    assert relerr(dq, qg.grad) < 1e-4 and relerr(dkv, kvg.grad) < 1e-4
    # patch folding = Conv2d(k=P, s=P) im2col
    img = torch.randn(2, 3, 32, 48)
    rows = ops.patchify(img.to(dev), 16)
    ref_rows = F.unfold(img, 16, stride=16).transpose(1, 2).reshape(-1, 3 * 256)
    assert relerr(rows, ref_rows) == 0
    assert relerr(ops.unpatchify(rows, None, 2, 3, 2, 3, 16), img) == 0
