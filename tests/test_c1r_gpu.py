"""csrc/c1r.hip — the register-direct 1x1 convolutions of the C <= 256 Bottleneck stages (pipeline/models/ae_64x8x8_lin.py:15,19),
through the C ABI: against torch fp64 on the CPU (the oracle's primitive for this op is F.conv2d), against the fp32 kernels
of gemm.hip on the same tensors, bit-identity of the fused BatchNorm + GELU prologue, and the BatchNorm sums of the epilogue."""
import pytest
import torch
import torch.nn.functional as F

from tests._util import relerr

pytestmark = pytest.mark.gpu

# (M, K) served: the C = 128 and C = 256 stages in both directions (one block owns all rows), and M-sliced (blocks of one XCD
# share the tiles and split the rows): the widening products of C = 512 / 1024
SHAPES = [(32, 128), (64, 256), (128, 32), (256, 64), (512, 128), (1024, 256)]
FWD_SHAPES = SHAPES


def rnd(shape, seed, lo=-1.0, hi=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(shape, generator=g) * (hi - lo) + lo).float()


@pytest.fixture(scope="module")
def ops(dev):
    from weatherforecastingtoolkit_amd import ops as o
    return o


def err64(y, ref64):
    return float((y.double().cpu() - ref64).abs().max() / ref64.abs().max())


@pytest.mark.parametrize("m,k", SHAPES)
@pytest.mark.parametrize("nb,h,w", [(1, 8, 8), (3, 16, 20), (2, 24, 24), (37, 8, 16), (32, 48, 48)])
def test_c1r_forward_and_data_gradient_match_fp64_like_the_fp32_kernels(ops, dev, m, k, nb, h, w):
    """forward (A = w) and data gradient (A = w^T) on c1r vs fp64; the error must stay at the level of gemm.hip's exact-fp32
    MFMA kernel on the same tensors.  Grids from one partial block (8 tiles) to several tiles per wave (37 images)."""
    assert ops.c1r_supported(m, k, h * w)
    x = rnd((nb, k, h, w), 1).to(dev)
    wt = rnd((m, k, 1, 1), 2, -0.3, 0.3).to(dev)           # forward weight (Cout = m, Cin = k)
    wt_t = rnd((k, m, 1, 1), 3, -0.3, 0.3).to(dev)         # a (Cout = k, Cin = m) weight whose data gradient is the same product
    ref_f = F.conv2d(x.double().cpu(), wt.double().cpu())
    ref_d = F.conv_transpose2d(x.double().cpu(), wt_t.double().cpu())
    res = {}
    ops.set_c1r(False)
    try:
        res[False] = (ops.conv1x1_fwd(x, wt), ops.conv1x1_bwd_data(x, wt_t))
    finally:
        ops.set_c1r(True)
    res[True] = (ops._c1r(wt, False, x, None, None, False, "test"), ops._c1r(wt_t, True, x, None, None, False, "test"))
    assert torch.equal(res[True][1], ops.conv1x1_bwd_data(x, wt_t)), "the data gradient of every served shape is routed to c1r"
    if (m, k) in FWD_SHAPES:
        assert torch.equal(res[True][0], ops.conv1x1_fwd(x, wt))
    assert not torch.equal(res[True][0], res[False][0]), "c1r did not run (same bits as gemm.hip)"
    for i, ref in enumerate((ref_f, ref_d)):
        e_old, e_new = err64(res[False][i], ref), err64(res[True][i], ref)
        assert e_new <= max(2.0 * e_old, 2e-6), (i, e_new, e_old)


@pytest.mark.parametrize("m,k", FWD_SHAPES)
def test_c1r_prologue_residual_and_stats(ops, dev, m, k):
    nb, h, w = 5, 16, 24
    x = rnd((nb, k, h, w), 4, -2.0, 2.0).to(dev)
    wt = rnd((m, k, 1, 1), 5, -0.3, 0.3).to(dev)
    r = rnd((nb, m, h, w), 6).to(dev) if m > k else None
    ones, zeros = torch.ones(k, device=dev), torch.zeros(k, device=dev)
    st = ops.bn_stats_train(x, rnd((k,), 7, 0.5, 1.5).to(dev), rnd((k,), 8).to(dev), zeros.clone(), ones.clone())
    a = ops.bn_act_fwd(x, st, 1)
    # (1) the fused BatchNorm + GELU prologue is bit-identical to the two-kernel form
    y_two = ops.conv1x1_fwd(a, wt, None, r)
    y_fused = ops.conv1x1_fwd_bnact(x, st, wt, None, r)
    assert torch.equal(y_two, y_fused)
    ref = F.conv2d(a.double().cpu(), wt.double().cpu()) + (0 if r is None else r.double().cpu())
    assert err64(y_fused, ref) < 3e-6
    # (2) BatchNorm sums of the epilogue == the separate statistics pass on the stored tensor; the result is unchanged
    gamma, beta = rnd((m,), 9, 0.5, 1.5).to(dev), rnd((m,), 10).to(dev)
    for fn in (lambda: ops.conv1x1_fwd_stats(a, wt, None, r), lambda: ops.conv1x1_fwd_bnact(x, st, wt, None, r, stats=True)):
        y, sr = fn()
        assert torch.equal(y, y_two) and sr is not None
        rm0, rv0, rm1, rv1 = torch.zeros(m, device=dev), torch.ones(m, device=dev), torch.zeros(m, device=dev), torch.ones(m, device=dev)
        s_sep = ops.bn_stats_train(y, gamma, beta, rm0, rv0)
        s_epi = ops.bn_stats_from_rows(sr, tuple(y.shape), gamma, beta, rm1, rv1)
        for name in ("mean", "invstd", "scale", "shift"):
            assert relerr(getattr(s_epi, name), getattr(s_sep, name)) < 2e-7, name
        assert relerr(rm1, rm0) < 2e-7 and relerr(rv1, rv0) < 2e-7
    # (3) repeat launches are bit-identical (fixed tile -> wave assignment, no atomics)
    y2, sr2 = ops.conv1x1_fwd_bnact(x, st, wt, None, r, stats=True)
    assert torch.equal(y2, y_two) and torch.equal(sr2.part, sr.part)


def test_c1r_full_size_stage_shapes(ops, dev):
    """the four products at B = 32 and the model's resolutions (384x384 with 128 channels, 192x192 with 256), where every wave
    walks MANY tiles (the cross-tile prefetch rings are live): against gemm.hip's exact-fp32 kernels on the same tensors —
    every element within fp32 rounding of a K-term dot product — in all the forms the step launches (prologue, residual,
    BatchNorm sums)"""
    for c, hh in ((128, 384), (256, 192), (512, 96), (1024, 48), (1024, 24)):
        mid = c // 4
        for m, k in ((mid, c), (c, mid)):
            if not ops.c1r_supported(m, k, hh * hh):
                continue
            x = torch.rand((32, k, hh, hh), device=dev) - 0.5
            wt = (torch.rand((m, k, 1, 1), device=dev) - 0.5) * 0.2
            wt_t = (torch.rand((k, m, 1, 1), device=dev) - 0.5) * 0.2
            r = (torch.rand((32, m, hh, hh), device=dev) - 0.5) if m > k else None
            st = ops.bn_stats_train(x, torch.ones(k, device=dev), torch.zeros(k, device=dev), torch.zeros(k, device=dev),
                                    torch.ones(k, device=dev))
            forms = [lambda: ops.conv1x1_fwd(x, wt, None, r), lambda: ops.conv1x1_fwd_bnact(x, st, wt, None, r),
                     lambda: ops.conv1x1_fwd_stats(x, wt, None, r)[0], lambda: ops.conv1x1_bwd_data(x, wt_t)]
            for i, fn in enumerate(forms):
                y_new = fn()
                ops.set_c1r(False)
                try:
                    y_old = fn()
                finally:
                    ops.set_c1r(True)
                if ops._c1r_take(m, k, hh * hh, i == 3):
                    assert not torch.equal(y_new, y_old), "c1r did not run"
                scale = float(y_old.abs().max())
                d = float((y_new - y_old).abs().max()) / scale
                assert d < 5e-6, (c, m, k, i, d)      # every element, the last tile of the last image included
                del y_new, y_old
            del x, r


@pytest.mark.parametrize("m,k", [(128, 32), (256, 64)])
@pytest.mark.parametrize("nb,h,w,res", [(2, 16, 16, False), (5, 16, 24, True), (40, 16, 16, True)])
def test_c1r_bnred_epilogue_equals_the_reduce_pass(ops, dev, m, k, nb, h, w, res):
    """wfae_c1r_bnred: the data gradient dA = W^T dT of the C -> C/4 convolution with phase 1 of the BatchNorm + GELU backward of the
    layer in front (sum dU, sum dU xhat) in its epilogue, finished by wfae_bn_act_bwd_from_rows + phase 2 — against the plain
    sequence data gradient -> wfae_bn_act_bwd (reduce pass + dx pass) on the same tensors: dA bit-identical to the plain c1r
    data gradient, dgamma / dbeta to fp64-sum accuracy, dx to the rounding of its two coefficients"""
    assert ops.c1r_bnred_supported(m, k, h * w)
    dt = rnd((nb, k, h, w), 1).to(dev)
    x = rnd((nb, m, h, w), 2, -2.0, 2.0).to(dev)
    wt = rnd((k, m, 1, 1), 3, -0.3, 0.3).to(dev)           # the (Cout = k, Cin = m) weight of the C -> C/4 convolution
    r = rnd((nb, m, h, w), 4).to(dev) if res else None
    gamma = rnd((m,), 5, 0.5, 1.5).to(dev)
    st = ops.bn_stats_train(x, gamma, rnd((m,), 6).to(dev), torch.zeros(m, device=dev), torch.ones(m, device=dev))
    da_ref = ops.conv1x1_bwd_data(dt, wt)
    dg0, db0 = torch.empty(m, device=dev), torch.empty(m, device=dev)
    dx_ref = ops.bn_act_bwd(da_ref, x, gamma, st, dg0, db0, r, 1, True)
    da, sr = ops.c1r_bnred(wt, dt, x, st)
    assert torch.equal(da, da_ref)
    dg1, db1 = torch.empty(m, device=dev), torch.empty(m, device=dev)
    ops.bn_act_bwd_from_rows(sr, m, dg1, db1)
    dx = ops.bn_act_bwd_dx(da, x, gamma, st, r, 1, True)
    assert relerr(dg1, dg0) < 1e-6 and relerr(db1, db0) < 1e-6
    assert relerr(dx, dx_ref) < 1e-6
    # accumulate form of the finalize, repeatability
    ops.bn_act_bwd_from_rows(sr, m, dg1, db1, accumulate=True)
    assert relerr(dg1, 2 * dg0) < 1e-6
    da2, sr2 = ops.c1r_bnred(wt, dt, x, st)
    assert torch.equal(da2, da) and torch.equal(sr2.part, sr.part)


@pytest.mark.parametrize("m,k", [(128, 32), (256, 64)])
@pytest.mark.parametrize("nb,h,w,res,training", [(2, 16, 16, False, True), (5, 16, 24, True, True), (40, 16, 16, True, True),
                                                 (3, 8, 24, True, False)])
def test_c1r_bndx_recomputed_gradient_equals_the_stored_one(ops, dev, m, k, nb, h, w, res, training):
    """wfae_c1r_bnred(da = NULL) + wfae_bn_act_bwd_from_rows + wfae_c1r_bndx (the second pass of the BatchNorm + GELU backward in
    the epilogue of the data gradient computed again) against the da-storing sequence on the same tensors: the same partial
    rows bit for bit, dx to the rounding of one fused multiply-add (both evaluate bn_act_bwd_dx_kernel's expression on a
    bit-identical dA), and against the plain wfae_conv1x1_bwd_data -> wfae_bn_act_bwd sequence"""
    assert ops.c1r_bnred_supported(m, k, h * w)
    dt = rnd((nb, k, h, w), 11).to(dev)
    x = rnd((nb, m, h, w), 12, -2.0, 2.0).to(dev)
    wt = rnd((k, m, 1, 1), 13, -0.3, 0.3).to(dev)
    r = rnd((nb, m, h, w), 14).to(dev) if res else None
    gamma = rnd((m,), 15, 0.5, 1.5).to(dev)
    st = ops.bn_stats_train(x, gamma, rnd((m,), 16).to(dev), torch.zeros(m, device=dev), torch.ones(m, device=dev))
    dg0, db0 = torch.empty(m, device=dev), torch.empty(m, device=dev)
    dx_plain = ops.bn_act_bwd(ops.conv1x1_bwd_data(dt, wt), x, gamma, st, dg0, db0, r, 1, training)
    da, sr = ops.c1r_bnred(wt, dt, x, st)
    dg1, db1 = torch.empty(m, device=dev), torch.empty(m, device=dev)
    ops.bn_act_bwd_from_rows(sr, m, dg1, db1)
    dx_stored = ops.bn_act_bwd_dx(da, x, gamma, st, r, 1, training)
    none, sr2 = ops.c1r_bnred(wt, dt, x, st, store=False)
    assert none is None and sr2.rows == sr.rows and torch.equal(sr2.part, sr.part)
    dg2, db2 = torch.empty(m, device=dev), torch.empty(m, device=dev)
    ops.bn_act_bwd_from_rows(sr2, m, dg2, db2)
    assert torch.equal(dg2, dg1) and torch.equal(db2, db1)
    dx = ops.c1r_bndx(wt, dt, x, gamma, st, r, training)
    scale = dx_stored.abs().max().item()
    assert (dx - dx_stored).abs().max().item() <= 4e-7 * scale
    assert relerr(dx, dx_plain) < 1e-6 and relerr(dg2, dg0) < 1e-6 and relerr(db2, db0) < 1e-6
    assert torch.equal(ops.c1r_bndx(wt, dt, x, gamma, st, r, training), dx)      # repeatable


def test_c1r_bndx_in_the_bottleneck_backward(ops, dev):
    """functional._dgrad_bn with the switch on and off: the same dx / dgamma / dbeta"""
    from weatherforecastingtoolkit_amd import functional as Fn
    m, k, nb, h, w = 128, 32, 6, 16, 32
    dt, x, dy = rnd((nb, k, h, w), 21).to(dev), rnd((nb, m, h, w), 22, -2.0, 2.0).to(dev), rnd((nb, m, h, w), 23).to(dev)
    wt, gamma = rnd((k, m, 1, 1), 24, -0.3, 0.3).to(dev), rnd((m,), 25, 0.5, 1.5).to(dev)
    st = ops.bn_stats_train(x, gamma, rnd((m,), 26).to(dev), torch.zeros(m, device=dev), torch.ones(m, device=dev))
    out = []
    try:
        for on in (False, True):
            ops.set_c1r_bndx(on)
            dg, db = torch.empty(m, device=dev), torch.empty(m, device=dev)
            out.append((Fn._dgrad_bn(dt, wt, None, x, gamma, st, dg, db, dy, True), dg, db))
    finally:
        ops.set_c1r_bndx(True)
    assert relerr(out[1][0], out[0][0]) < 1e-6 and torch.equal(out[1][1], out[0][1]) and torch.equal(out[1][2], out[0][2])


@pytest.mark.parametrize("m,k,nb,h,w", [(32, 128, 40, 16, 16), (64, 256, 8, 16, 24), (128, 32, 40, 16, 16)])
def test_c1r_prologue_is_repeatable_with_cold_operands(ops, dev, m, k, nb, h, w):
    """the fp32 prologue forms under the stress that exposed csrc/c1rb.hip's two-waves-per-SIMD failure: 60 launches, caches
    emptied in front of each, one result"""
    x = rnd((nb, k, h, w), 1, -2.0, 2.0).to(dev)
    wt = (rnd((m, k, 1, 1), 3) * k ** -0.5).to(dev)
    st = ops.BnStats(k, dev)
    st.scale.copy_(rnd((k,), 5) + 1.5)
    st.shift.copy_(rnd((k,), 6))
    assert ops.c1r_supported(m, k, h * w)
    junk = torch.zeros(256 << 20, dtype=torch.uint8, device=dev)
    first = ops.conv1x1_fwd_bnact(x, st, wt)
    for _ in range(60):
        junk.add_(1)
        assert torch.equal(ops.conv1x1_fwd_bnact(x, st, wt), first)

