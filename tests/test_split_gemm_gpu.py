"""GPU: the split-operand GEMMs (csrc/splitgemm.hip) — fp32 values as three bf16 planes, six bf16 MFMA products with
fp32 accumulation.  The claim under test is ACCURACY: the result is as close to the fp64 product as the fp32-MFMA
kernels' (and torch's fp32 matmul), so the default fp32 path may run on the bf16 matrix pipe."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops(dev):
    from weatherforecastingtoolkit_amd import ops as o
    return o


def _planes_to_f32(p3):
    """(3, ...) int16 bf16 bit patterns -> three fp32 tensors"""
    return [(p.to(torch.int32) << 16).view(torch.float32) for p in p3]


def test_split_is_exact(ops, dev):
    """x == h + m + l bit for bit, over 40 binades, both signs, including values whose bf16 rounding carries"""
    g = torch.Generator().manual_seed(0)
    x = (torch.rand(1 << 20, generator=g) - 0.5) * torch.exp2(torch.randint(-20, 20, (1 << 20,), generator=g).float())
    x[:4] = torch.tensor([0.0, 1.0, -1.0, 1.9999999])
    x = x.to(dev)
    h, m, l = _planes_to_f32(ops.split_bf16x3(x))
    assert torch.equal((h + m) + l, x)
    assert float((m.abs() > h.abs() * 2.0 ** -8).sum()) == 0 and float((l.abs() > h.abs() * 2.0 ** -16).sum()) == 0


def _err(c, ref64, scale64):
    return float(((c.double() - ref64).abs() / scale64).max())


@pytest.mark.parametrize("kind", [0, 1])
@pytest.mark.parametrize("y,m,n,k", [(1, 256, 128, 32), (3, 300, 200, 96), (2, 64, 136, 64), (1, 512, 384, 2048), (2, 40, 8, 32), (9, 300, 200, 64),
                                     (40, 300, 512, 128), (25, 512, 704, 64), (70, 256, 384, 256)])
def test_split_gemm_matches_fp32_accuracy(ops, dev, kind, y, m, n, k):
    """C = A B from split operands against fp64: the error (relative to sum |a||b|, the natural scale of a dot product)
    is within 1.5x of torch's own fp32 matmul on the same data, at every tile tail (M, N not multiples of the 256 x 128
    tile, several batches).  The last three shapes have more tiles than the chip has CUs and an even number of K-steps: the
    persistent launch, whose loader fetches a workgroup's next tile while the current one is finished (2, 4 and 8 K-steps,
    partial tiles among them)"""
    g = torch.Generator().manual_seed(1000 * kind + m + n + k)
    a = (torch.randn(y, m, k, generator=g) * torch.exp2(torch.randint(-6, 6, (y, m, 1), generator=g).float())).to(dev)
    b = torch.randn(y, k, n, generator=g).to(dev)
    ref = a.double() @ b.double()
    scale = a.double().abs() @ b.double().abs()
    a3 = ops.split_bf16x3(a)
    b3 = ops.split_bf16x3(b if kind == 0 else b.transpose(1, 2).contiguous())
    c = ops.split_gemm(a3, b3, kind)
    e_split, e_f32 = _err(c, ref, scale), _err(a @ b, ref, scale)
    assert e_split <= 1.5 * e_f32 + 3e-8, (e_split, e_f32)
    assert e_split < 4e-7


@pytest.mark.parametrize("mode,nb,chi,clo,hlo,wlo", [("f42", 2, 64, 128, 16, 16), ("f42", 4, 32, 64, 16, 24), ("f22", 2, 16, 32, 16, 16),
                                                      ("f42", 8, 256, 512, 24, 24), ("f22", 8, 8, 32, 4, 4)])
def test_winograd_on_split_gemms_matches_fp32_mfma_accuracy(ops, dev, mode, nb, chi, clo, hlo, wlo):
    """the three 4x4 stride-2 products through the split GEMMs: against an fp64 convolution their error is that of the
    fp32-MFMA Winograd path (within 1.5x), and the split entry points are the ones that ran"""
    g = torch.Generator().manual_seed(7)
    hi = (torch.rand(nb, chi, 2 * hlo, 2 * wlo, generator=g) - 0.5).to(dev)
    lo = (torch.rand(nb, clo, hlo, wlo, generator=g) - 0.5).to(dev)
    w = ((torch.rand(clo, chi, 4, 4, generator=g) - 0.5) * 0.2).to(dev)
    hr, wr = hi.double().requires_grad_(True), w.double().requires_grad_(True)
    ref = F.conv2d(hr, wr, stride=2, padding=1)
    ref.backward(lo.double())
    res = {}
    ops.set_winograd(mode)
    try:
        for split in (False, True):
            ops.set_split_gemm(split)
            pl = ops.wino_plan(nb, chi, clo, hlo, wlo)
            assert pl is not None and pl.split == split
            dw = torch.empty_like(w)
            ops.conv4x4s2_wgrad(lo, hi, dw)
            res[split] = (ops.conv4x4s2_down(hi, w), ops.conv4x4s2_up(lo, w), dw)
    finally:
        ops.set_winograd("auto")
        ops.set_split_gemm(True)
    for i, r64 in enumerate((ref, hr.grad, wr.grad)):
        r64 = r64.detach()
        rms = float(r64.pow(2).mean().sqrt())
        e_f32 = float((res[False][i].double() - r64).abs().max()) / rms
        e_split = float((res[True][i].double() - r64).abs().max()) / rms
        assert e_split <= 1.5 * e_f32 + 1e-7, (i, e_split, e_f32)


def test_split_gemm_refuses_unserved_shapes(ops, dev):
    from weatherforecastingtoolkit_amd import _lib
    a3 = torch.zeros(3, 1, 32, 48, dtype=torch.int16, device=dev)   # K = 48 is not a multiple of 32
    b3 = torch.zeros(3, 1, 48, 64, dtype=torch.int16, device=dev)
    with pytest.raises(_lib.WfaeError):
        ops.split_gemm(a3, b3, 0)
    assert _lib.load().wfae_wino_split_supported(1, 1, 256, 512, 8, 8) == 0    # T = 4 tiles
    assert _lib.load().wfae_wino_split_supported(1, 32, 256, 512, 96, 96) == 1


@pytest.mark.parametrize("n", [200, 512, 600])
@pytest.mark.parametrize("kind", [0, 1])
def test_one_plane_gemm_is_the_bf16_operand_product(ops, dev, kind, n):
    """planes = 1: the h plane alone — the product of the bf16-rounded operands with fp32 accumulation ('medium'
    precision with 2-byte operand storage): equal to an fp64 product of the rounded operands to fp32 accumulation error.
    n >= 256 runs the 256-column blocks (64 x 128 wave tiles), 600 with a partial last block"""
    g = torch.Generator().manual_seed(11 + kind)
    a = torch.randn(3, 300, 96, generator=g).to(dev)
    b = torch.randn(3, 96, n, generator=g).to(dev)
    a1, b1 = ops.split_bf16x3(a, planes=1), ops.split_bf16x3(b if kind == 0 else b.transpose(1, 2).contiguous(), planes=1)
    assert a1.shape[0] == 1 and torch.equal(_planes_to_f32(a1)[0], a.bfloat16().float())
    c = ops.split_gemm(a1, b1, kind)
    ar, br = a.bfloat16().double(), b.bfloat16().double()
    ref, scale = ar @ br, ar.abs() @ br.abs()
    assert _err(c, ref, scale) < 2e-7


def test_winograd_medium_precision_on_one_plane_operands(ops, dev):
    """'medium' matmul precision: the Winograd products read 2-byte operands (the h plane written by the transforms) and
    agree with the fp32-storage bf16 path (operands rounded behind the LDS read) — the same rounded values, another
    accumulation order"""
    g = torch.Generator().manual_seed(5)
    nb, chi, clo, hlo = 4, 64, 128, 16
    hi = (torch.rand(nb, chi, 2 * hlo, 2 * hlo, generator=g) - 0.5).to(dev)
    lo = (torch.rand(nb, clo, hlo, hlo, generator=g) - 0.5).to(dev)
    w = ((torch.rand(clo, chi, 4, 4, generator=g) - 0.5) * 0.2).to(dev)
    res = {}
    ops.set_float32_matmul_precision("medium")
    try:
        for split in (False, True):
            ops.set_split_gemm(split)
            pl = ops.wino_plan(nb, chi, clo, hlo, hlo)
            assert pl is not None and pl.split == split and (not split or pl.planes == 1)
            dw = torch.empty_like(w)
            ops.conv4x4s2_wgrad(lo, hi, dw)
            res[split] = (ops.conv4x4s2_down(hi, w), ops.conv4x4s2_up(lo, w), dw)
    finally:
        ops.set_float32_matmul_precision("highest")
        ops.set_split_gemm(True)
    for a, b in zip(res[False], res[True]):
        assert float((a - b).abs().max() / a.abs().max()) < 2e-6


@pytest.mark.parametrize("nb,cin,cout,h", [(16, 512, 128, 48), (16, 128, 512, 48), (32, 1024, 256, 24), (32, 256, 1024, 24)])
def test_conv1x1_in_register_split_matches_fp32_accuracy(ops, dev, nb, cin, cout, h):
    """the generic GEMM kernel with the operands split in registers (gemm_kernel PREC 2: K >= 128, M >= 64): forward, data
    gradient and weight gradient of the 1x1 convolution against fp64 — within 1.5x of the fp32-MFMA kernel's error"""
    g = torch.Generator().manual_seed(cin + cout)
    x = (torch.rand(nb, cin, h, h, generator=g) - 0.5).to(dev)
    w = ((torch.rand(cout, cin, 1, 1, generator=g) - 0.5) * 0.1).to(dev)
    dy = (torch.rand(nb, cout, h, h, generator=g) - 0.5).to(dev)
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    ref = F.conv2d(xr, wr)
    ref.backward(dy.double())
    res = {}
    try:
        for split in (False, True):
            ops.set_split_gemm(split)
            dw = torch.empty_like(w)
            ops.conv1x1_bwd_weight(dy, x, dw)
            res[split] = (ops.conv1x1_fwd(x, w), ops.conv1x1_bwd_data(dy, w), dw)
    finally:
        ops.set_split_gemm(True)
    for i, r64 in enumerate((ref.detach(), xr.grad, wr.grad)):
        rms = float(r64.pow(2).mean().sqrt())
        e_f32 = float((res[False][i].double() - r64).abs().max()) / rms
        e_split = float((res[True][i].double() - r64).abs().max()) / rms
        assert e_split <= 1.5 * e_f32 + 1e-7, (i, e_split, e_f32)
        assert not torch.equal(res[False][i], res[True][i]), "the split path did not run"


@pytest.mark.parametrize("nb,c,h,w", [(2, 128, 48, 96), (1, 128, 96, 96), (3, 256, 48, 48), (2, 256, 16, 48), (2, 256, 24, 24),
                                      (1, 256, 12, 24)])
def test_grouped3x3_weight_gradient_on_split_planes_matches_fp32_accuracy(ops, dev, nb, c, h, w):
    """csrc/g3b.hip g3bw_kernel on fp32 tensors (three exact bf16 planes per value, six products per fp32 product) against
    fp64, beside the fp32-MFMA kernel of dconv.hip on the same data: inside the spread two fp32 summation orders of these
    10^4 .. 10^5-term sums show (3x, measured 0.5 - 2x); accumulate form; the shapes of the model that the kernel serves in
    fp32 (W <= 96), several strips and slabs"""
    groups = 8
    g = torch.Generator().manual_seed(c + w)
    x = (torch.rand(nb, c, h, w, generator=g) - 0.5).to(dev)
    dy = (torch.rand(nb, c, h, w, generator=g) - 0.5).to(dev)
    wr = torch.zeros(c, c // groups, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double().cpu(), wr, padding=1, groups=groups).backward(dy.double().cpu())
    r64 = wr.grad.to(dev)
    res = {}
    try:
        for on in (False, True):
            ops.set_g3b(on)
            dw = torch.empty(c, c // groups, 3, 3, device=dev)
            ops.gconv3x3_bwd_weight(dy, x, dw, groups)
            res[on] = dw
        dw2 = res[True].clone()
        ops.gconv3x3_bwd_weight(dy, x, dw2, groups, accumulate=True)
    finally:
        ops.set_g3b(True)
    rms = float(r64.pow(2).mean().sqrt())
    e_old = float((res[False].double() - r64).abs().max()) / rms
    e_new = float((res[True].double() - r64).abs().max()) / rms
    assert e_new <= max(3.0 * e_old, 3e-6), (e_new, e_old)
    assert not torch.equal(res[False], res[True]), "the split-plane kernel did not run"
    assert float((dw2 - 2 * res[True]).abs().max()) <= 1e-6 * float(res[True].abs().max())


@pytest.mark.parametrize("nb,c,h,w", [(2, 128, 48, 96), (1, 128, 96, 96)])
def test_grouped3x3_forward_on_split_planes_matches_fp32_accuracy(ops, dev, nb, c, h, w):
    """csrc/g3b.hip g3b_kernel on fp32 tensors (x and the weights as three exact bf16 planes, six products per fp32 product,
    tap shifts on the output side) against fp64, forward and data gradient, beside the dconv.hip kernels on the same data"""
    groups = 8
    g = torch.Generator().manual_seed(c + h)
    x = (torch.rand(nb, c, h, w, generator=g) - 0.5).to(dev)
    wt = ((torch.rand(c, c // groups, 3, 3, generator=g) - 0.5) * 0.3).to(dev)
    dy = (torch.rand(nb, c, h, w, generator=g) - 0.5).to(dev)
    xr = x.double().cpu().requires_grad_(True)
    ref = F.conv2d(xr, wt.double().cpu(), padding=1, groups=groups)
    ref.backward(dy.double().cpu())
    refs = (ref.detach().to(dev), xr.grad.to(dev))
    res = {}
    from weatherforecastingtoolkit_amd import _lib

    def direct(t, tr):   # the entry point itself
        out, ws = torch.empty_like(t), ops.workspace()
        _lib.call("wfae_g3b_fwd", t.data_ptr(), wt.data_ptr(), out.data_ptr(), nb, c, h, w, groups, int(tr), ws.data_ptr(),
                  ws.numel(), torch.cuda.current_stream().cuda_stream)
        return out

    ops.set_g3b(False)
    try:
        res[False] = (ops.gconv3x3_fwd(x, wt, groups, False), ops.gconv3x3_fwd(dy, wt, groups, True))
    finally:
        ops.set_g3b(True)
    res[True] = (direct(x, False), direct(dy, True))
    if w == 96:
        assert torch.equal(res[True][0], ops.gconv3x3_fwd(x, wt, groups, False)), "16 channels per group @96 is routed to g3b"
    for i, r64 in enumerate(refs):
        rms = float(r64.pow(2).mean().sqrt())
        e_old = float((res[False][i].double() - r64).abs().max()) / rms
        e_new = float((res[True][i].double() - r64).abs().max()) / rms
        assert e_new <= max(2.0 * e_old, 1e-6), (i, e_new, e_old)
        assert not torch.equal(res[False][i], res[True][i]), "the split-plane kernel did not run"
