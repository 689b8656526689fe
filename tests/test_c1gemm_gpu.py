"""GPU parity of csrc/c1gemm.hip — the Bottleneck's 1x1 convolutions (reference pipeline/models/ae_64x8x8_lin.py:15,19) on
the bf16 matrix pipe with exact three-plane operands, and the BatchNorm + GELU backward fused into their data gradient —
against fp64 torch on the CPU and against the kernels they replace (wfae_conv1x1_*, wfae_bn_act_bwd)."""
import numpy as np
import pytest
import torch

from tests._util import relerr

pytestmark = pytest.mark.gpu


def _rnd(gen, *shape, scale=1.0):
    return (torch.rand(shape, generator=gen) * 2 - 1) * scale


# (NB, K, M, H, W): the three block tiles (M % 256 / % 128 / % 64), one K-step and many, tiles that cross image
# boundaries, column tails (N not a multiple of the tile width), M tails of the 256-row tile
SHAPES = [(2, 64, 256, 24, 24), (3, 128, 128, 10, 10), (2, 256, 64, 16, 16), (1, 32, 128, 8, 4), (2, 96, 512, 12, 12),
          (5, 32, 64, 6, 6), (2, 1024, 256, 8, 8), (1, 64, 320, 20, 20)]


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("mode", ["plain", "res", "pro", "res+stats", "pro+stats"])
def test_c1gemm_fwd_matches_fp64(dev, shape, mode):
    from weatherforecastingtoolkit_amd import ops
    nb, k, m, h, w = shape
    assert ops.c1gemm_supported(m, k, h * w)
    gen = torch.Generator().manual_seed(hash((shape, mode)) % (2 ** 31))
    x, wt = _rnd(gen, nb, k, h, w, scale=2.0), _rnd(gen, m, k, scale=k ** -0.5)
    res = _rnd(gen, nb, m, h, w) if "res" in mode else None
    st = None
    xin = x.double()
    if "pro" in mode:
        st = ops.BnStats(k, dev)
        sc, sh = _rnd(gen, k) + 1.5, _rnd(gen, k)
        st.scale.copy_(sc)
        st.shift.copy_(sh)
        xin = torch.nn.functional.gelu(x.double() * sc.double().view(1, k, 1, 1) + sh.double().view(1, k, 1, 1))
    want = torch.einsum("mk,nkhw->nmhw", wt.double(), xin)
    mag = torch.einsum("mk,nkhw->nmhw", wt.double().abs(), xin.abs()).max().item()     # sum |a||b|: the error scale
    if res is not None:
        want = want + res.double()
    W3, Wt3 = ops.c1_split_weights(wt.to(dev))
    # the planes reproduce the weight exactly (h + m + l == w) and Wt3 is the transpose of W3
    def planes_to_f32(P):
        return (P.to(torch.int32) << 16).view(torch.float32).double().sum(0)
    assert torch.equal(planes_to_f32(W3.cpu()).float(), wt) and torch.equal(Wt3.cpu(), W3.cpu().transpose(1, 2))
    out = ops.c1gemm_fwd(W3, x.to(dev), st, None if res is None else res.to(dev), "stats" in mode)
    y, sr = out if "stats" in mode else (out, None)
    err = (y.double().cpu() - want).abs().max().item() / mag
    assert err < 3e-7, err                      # fp32 accumulation of exact products: the fp32 GEMM's own error level
    if "pro" not in mode:                       # and it tracks the kernel it replaces
        old = ops.conv1x1_fwd(x.to(dev), wt.to(dev), None, None if res is None else res.to(dev))
        assert relerr(y, old) < 2e-5
    if sr is not None:
        bn = torch.nn.BatchNorm2d(m).to(dev)
        st_a = ops.bn_stats_from_rows(sr, tuple(y.shape), bn.weight, bn.bias, None, None)
        st_b = ops.bn_stats_train(y, bn.weight, bn.bias, torch.zeros(m, device=dev), torch.ones(m, device=dev))
        assert relerr(st_a.mean, st_b.mean) < 2e-6 and relerr(st_a.invstd, st_b.invstd) < 2e-6


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("training", [True, False])
def test_c1gemm_bn_backward_epilogues(dev, shape, training):
    """dA = W^T dT feeding the BatchNorm + GELU backward of the layer in front: (a) reduce in the GEMM epilogue + the dx
    kernel, (b) reduce-only GEMM + dx recomputed in a second GEMM, both against the unfused chain
    wfae_conv1x1_bwd_data -> wfae_bn_act_bwd and against fp64 autograd"""
    from weatherforecastingtoolkit_amd import ops
    nb, k, m, h, w = shape
    gen = torch.Generator().manual_seed(1 + hash(shape) % (2 ** 31))
    dt, wt = _rnd(gen, nb, k, h, w), _rnd(gen, k, m, scale=k ** -0.5)      # conv weight (Cout = k, Cin = m): da = wt^T dt
    x, res = _rnd(gen, nb, m, h, w, scale=2.0) + 0.3, _rnd(gen, nb, m, h, w)
    gamma, beta = (_rnd(gen, m) + 1.5).to(dev), _rnd(gen, m).to(dev)
    xd, dtd, resd, wd = x.to(dev), dt.to(dev), res.to(dev), wt.to(dev)
    rm, rv = torch.zeros(m, device=dev), torch.ones(m, device=dev)
    if training:
        st = ops.bn_stats_train(xd, gamma, beta, rm, rv)
    else:
        rv = (_rnd(gen, m).abs() + 0.5).to(dev)
        rm = _rnd(gen, m).to(dev)
        st = ops.bn_fold_eval(gamma, beta, rm, rv)
    # unfused chain
    da0 = ops.conv1x1_bwd_data(dtd, wd)
    dg0, db0 = torch.empty(m, device=dev), torch.empty(m, device=dev)
    dx0 = ops.bn_act_bwd(da0, xd, gamma, st, dg0, db0, resd, 1, training)
    _, Wt3 = ops.c1_split_weights(wd)
    assert tuple(Wt3.shape) == (3, m, k)
    # (a) reduce in the epilogue, da stored
    da1, sr = ops.c1gemm_bnred(Wt3, dtd, xd, st)
    dg1, db1 = torch.empty(m, device=dev), torch.empty(m, device=dev)
    ops.bn_act_bwd_from_rows(sr, m, dg1, db1)
    dx1 = ops.bn_act_bwd_dx(da1, xd, gamma, st, resd, 1, training)
    assert relerr(da1, da0) < 2e-5
    sc = max(dg0.abs().max().item(), db0.abs().max().item())
    assert (dg1 - dg0).abs().max().item() < 3e-5 * sc and (db1 - db0).abs().max().item() < 3e-5 * sc
    assert relerr(dx1, dx0) < 3e-5
    # (b) da never stored
    none, sr2 = ops.c1gemm_bnred(Wt3, dtd, xd, st, store=False)
    assert none is None
    dg2, db2 = torch.empty(m, device=dev), torch.empty(m, device=dev)
    ops.bn_act_bwd_from_rows(sr2, m, dg2, db2)
    dx2 = ops.c1gemm_bndx(Wt3, dtd, xd, gamma, st, resd, training)
    assert torch.equal(dg2, dg1) and torch.equal(db2, db1)        # the same reduction, with or without the store
    assert relerr(dx2, dx0) < 3e-5
    # the same two fusions on gemm.hip's kernel (wfae_conv1x1_bwd_data_bnred / _bndx): any channel count % 4
    da3, sr3 = ops.conv1x1_bwd_data_bnred(dtd, wd, xd, st)
    dg3, db3 = torch.empty(m, device=dev), torch.empty(m, device=dev)
    ops.bn_act_bwd_from_rows(sr3, m, dg3, db3)
    dx3 = ops.bn_act_bwd_dx(da3, xd, gamma, st, resd, 1, training)
    assert relerr(da3, da0) < 2e-5      # the same GEMM up to the tile height (small grids: 64-row tiles for the epilogue)
    assert (dg3 - dg0).abs().max().item() < 3e-5 * sc and (db3 - db0).abs().max().item() < 3e-5 * sc
    assert relerr(dx3, dx0) < 3e-5
    none, sr4 = ops.conv1x1_bwd_data_bnred(dtd, wd, xd, st, store=False)
    assert none is None
    dg4, db4 = torch.empty(m, device=dev), torch.empty(m, device=dev)
    ops.bn_act_bwd_from_rows(sr4, m, dg4, db4)
    dx4 = ops.conv1x1_bwd_data_bndx(dtd, wd, xd, gamma, st, resd, training)
    assert torch.equal(dg4, dg3) and torch.equal(db4, db3)
    assert relerr(dx4, dx0) < 3e-5
    # fp64 autograd of  res-branch + conv1x1(gelu(bn(x)))  for the training-mode case
    if training:
        x64 = x.double().requires_grad_(True)
        g64, b64 = gamma.double().cpu().requires_grad_(True), beta.double().cpu().requires_grad_(True)
        a = torch.nn.functional.gelu(torch.nn.functional.batch_norm(x64, None, None, g64, b64, True, 0.0, 1e-5))
        t = torch.einsum("km,nmhw->nkhw", wt.double(), a)
        (t * dt.double()).sum().backward()
        assert relerr(dx2, x64.grad + res.double()) < 3e-5
        assert relerr(dg2, g64.grad) < 3e-5 and relerr(db2, b64.grad) < 3e-5


@pytest.mark.parametrize("shape", [(2, 128, 32, 12, 12), (3, 24, 20, 6, 6), (1, 16, 8, 4, 4), (2, 40, 132, 10, 10)])
def test_fused_bn_backward_on_the_fp32_gemm_kernel(dev, shape):
    """wfae_conv1x1_bwd_data_bnred / _bndx at channel counts c1gemm does not serve (Cin = 32: the C/4-wide data gradient of
    the C = 128 stage; odd multiples of 4; M tails): against the unfused chain"""
    from weatherforecastingtoolkit_amd import ops
    nb, k, m, h, w = shape
    assert ops.conv1x1_bn_fusable(m, h * w)
    gen = torch.Generator().manual_seed(7 + hash(shape) % (2 ** 31))
    dt, wt = _rnd(gen, nb, k, h, w).to(dev), _rnd(gen, k, m, scale=k ** -0.5).to(dev)
    x, res = (_rnd(gen, nb, m, h, w, scale=2.0) + 0.3).to(dev), _rnd(gen, nb, m, h, w).to(dev)
    gamma, beta = (_rnd(gen, m) + 1.5).to(dev), _rnd(gen, m).to(dev)
    st = ops.bn_stats_train(x, gamma, beta, torch.zeros(m, device=dev), torch.ones(m, device=dev))
    da0 = ops.conv1x1_bwd_data(dt, wt)
    dg0, db0 = torch.empty(m, device=dev), torch.empty(m, device=dev)
    dx0 = ops.bn_act_bwd(da0, x, gamma, st, dg0, db0, res, 1, True)
    sc = max(dg0.abs().max().item(), db0.abs().max().item())
    for store in (True, False):
        da, sr = ops.conv1x1_bwd_data_bnred(dt, wt, x, st, store=store)
        dg, db = torch.empty(m, device=dev), torch.empty(m, device=dev)
        ops.bn_act_bwd_from_rows(sr, m, dg, db)
        dx = ops.bn_act_bwd_dx(da, x, gamma, st, res, 1, True) if store else ops.conv1x1_bwd_data_bndx(dt, wt, x, gamma, st, res, True)
        assert (da is None) == (not store) and (da is None or relerr(da, da0) < 2e-5)
        assert (dg - dg0).abs().max().item() < 3e-5 * sc and (db - db0).abs().max().item() < 3e-5 * sc
        assert relerr(dx, dx0) < 3e-5


def test_bottleneck_paths_agree(dev):
    """one Bottleneck (C = 256 at 32 x 32: every 1x1 product of it is served by c1gemm) forward + backward on the c1gemm
    path, with the recompute form of the first BatchNorm's backward, and on the round-2 kernels: same results to fp32
    accumulation-order rounding; the reference's own goldens cover the absolute level (tests/test_model_gpu.py)"""
    from weatherforecastingtoolkit_amd import functional as Fn, ops
    from weatherforecastingtoolkit_amd.pipeline.models.ae_64x8x8_lin import Bottleneck
    torch.manual_seed(3)
    blk = Bottleneck(256).to(dev).train()
    with torch.no_grad():
        for p in blk.parameters():
            if p.dim() == 1:
                p.uniform_(0.5, 1.5)
    x0 = torch.randn(4, 256, 32, 32, device=dev)
    dy = torch.randn(4, 256, 32, 32, device=dev)
    outs = {}
    # (c1gemm on, its minimum K, fused BatchNorm-backward epilogues, recompute form)
    cfgs = {"round2": (False, 512, False, 0), "fused": (False, 512, True, 0), "fused_recompute": (False, 512, True, 1 << 30),
            "c1": (True, 32, True, 0), "c1_recompute": (True, 32, True, 1 << 30), "c1_unfused": (True, 32, False, 0)}
    keep = (ops._C1GEMM, ops._C1_MIN_K, Fn.C1_BNRED, Fn.C1_RECOMPUTE_MAXC)
    try:
        for name, (on, mink, red, rec) in cfgs.items():
            ops.set_c1gemm(on)
            ops._C1_MIN_K, Fn.C1_BNRED, Fn.C1_RECOMPUTE_MAXC = mink, red, rec
            for p in blk.parameters():
                p.grad = None
            x = x0.clone().requires_grad_(True)
            y = blk(x)
            y.backward(dy)
            outs[name] = [y.detach().clone(), x.grad.clone()] + [p.grad.clone() for p in blk.parameters()]
    finally:
        ops.set_c1gemm(keep[0])
        ops._C1_MIN_K, Fn.C1_BNRED, Fn.C1_RECOMPUTE_MAXC = keep[1:]
    for name in cfgs:
        for a, b in zip(outs[name], outs["round2"]):
            assert relerr(a, b) < 5e-5, name


@pytest.mark.parametrize("shape", [(2, 256, 64, 16, 16), (3, 64, 256, 8, 8), (2, 512, 128, 8, 8), (1, 128, 32, 16, 32),
                                   (2, 32, 128, 8, 8), (4, 1024, 256, 8, 4), (2, 160, 96, 8, 8), (1, 64, 64, 40, 40)])
@pytest.mark.parametrize("pro", [False, True])
def test_c1w_weight_gradient_fp32(dev, shape, pro):
    """csrc/c1w.hip (fp32 tensors, exact bf16x3 split at the LDS store, both operands as K-contiguous rows) against fp64 and
    against gemm.hip's weight gradient: every tile shape, either operand as the wide side (transposed slab reduce), the
    BatchNorm + GELU prologue on either side, split-K over images and pixel chunks, accumulate"""
    from weatherforecastingtoolkit_amd import ops
    nb, cin, cout, h, w = shape
    gen = torch.Generator().manual_seed(11 + hash(shape) % (2 ** 31))
    dy, x = _rnd(gen, nb, cout, h, w).to(dev), _rnd(gen, nb, cin, h, w, scale=2.0).to(dev)
    st = None
    xin = x.double().cpu()
    if pro:
        st = ops.BnStats(cin, dev)
        st.scale.copy_(_rnd(gen, cin) + 1.5)
        st.shift.copy_(_rnd(gen, cin))
        xin = torch.nn.functional.gelu(xin * st.scale.double().cpu().view(1, cin, 1, 1) + st.shift.double().cpu().view(1, cin, 1, 1))
    want = torch.einsum("nohw,nihw->oi", dy.double().cpu(), xin)
    mag = torch.einsum("nohw,nihw->oi", dy.double().cpu().abs(), xin.abs()).max().item()
    assert ops._c1w_route(dy, x, "")
    dw = torch.full((cout, cin, 1, 1), 7.0, device=dev)
    (ops.conv1x1_bwd_weight_bnact(dy, x, st, dw) if pro else ops.conv1x1_bwd_weight(dy, x, dw))
    err = (dw.view(cout, cin).double().cpu() - want).abs().max().item() / mag
    assert err < 3e-7, err
    ops.set_c1w(False)
    try:
        old = torch.empty_like(dw)
        (ops.conv1x1_bwd_weight_bnact(dy, x, st, old) if pro else ops.conv1x1_bwd_weight(dy, x, old))
    finally:
        ops.set_c1w(True)
    assert relerr(dw, old) < 2e-5
    # accumulate
    acc = dw.clone()
    (ops.conv1x1_bwd_weight_bnact(dy, x, st, acc, True) if pro else ops.conv1x1_bwd_weight(dy, x, acc, True))
    assert relerr(acc, 2 * dw) < 1e-6
    # bit-identical repeat (fixed-order slab reduce)
    again = torch.empty_like(dw)
    (ops.conv1x1_bwd_weight_bnact(dy, x, st, again) if pro else ops.conv1x1_bwd_weight(dy, x, again))
    assert torch.equal(again, dw)
