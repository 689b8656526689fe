"""Tensor-level launch wrappers over the libwfae.so C ABI (no autograd here).

torch is used for device memory, the current HIP stream and nothing else:
every arithmetic operation below is a hand-written gfx950 kernel reached
through ctypes.  All tensors must be CUDA(HIP) fp32 and contiguous (NCHW).
"""
from __future__ import annotations

import os

import torch

from . import _lib

_WS_DEFAULT = int(os.environ.get("WFAE_WORKSPACE_MB", "512")) << 20
_ws = {}


def _stream():
    return torch.cuda.current_stream().cuda_stream


# Optional per-launch timing with events on the launch stream (bench.py's
# roofline read-out): name -> [(start_evt, end_evt, algorithmic flops, algorithmic bytes)]
_prof = None


def profile_start():
    global _prof
    _prof = {}


PEAK_HBM = 8.0e12          # MI355X_MICROARCH.md
PEAK_FP32_MFMA = 157.3e12  # v_mfma_f32_32x32x2_f32 = the fp32 vector rate
PEAK_BF16_MFMA = 2500.0e12 # dense bf16


def profile_stop():
    """-> {name: (calls, total_ms, flops, bytes, ideal_ms, mfma_ms, hbm_ms)}; synchronises the device.  ideal_ms = sum over
    the launches of max(algorithmic bytes / HBM peak, algorithmic FLOPs / peak of the matrix instruction that launch runs on);
    mfma_ms / hbm_ms: the two terms summed on their own (which roof binds the label's launches as a whole)"""
    global _prof
    rec, _prof = _prof, None
    torch.cuda.synchronize()
    out = {}
    for k, lst in (rec or {}).items():
        ms = sum(a.elapsed_time(b) for a, b, _, _, _ in lst)
        out[k] = (len(lst), ms, sum(f for _, _, f, _, _ in lst), sum(b for _, _, _, b, _ in lst),
                  1e3 * sum(max(b / PEAK_HBM, f / pk) for _, _, f, b, pk in lst),
                  1e3 * sum(f / pk for _, _, f, _, pk in lst), 1e3 * sum(b / PEAK_HBM for _, _, _, b, _ in lst))
    return out


def _default_peak():
    return PEAK_BF16_MFMA if _lib.load().wfae_get_matmul_precision() == 1 else PEAK_FP32_MFMA


def _gemm_peak(m, k):
    """FLOP/s ceiling of the instruction a plain GEMM-family launch (1x1 / linear) runs on: bf16 operands at 'medium'; at
    fp32 precision the exact three-plane split (six bf16 products per fp32 product) for K >= 128, M >= 64 (gemm.hip
    launch_gemm_v), else v_mfma_f32_32x32x2_f32.  Work is counted in fp32-equivalent FLOPs."""
    if _lib.load().wfae_get_matmul_precision() == 1:
        return PEAK_BF16_MFMA
    if _SPLIT_GEMM and k >= 128 and m >= 64:
        return PEAK_BF16_MFMA / 6
    return PEAK_FP32_MFMA


def _call(name, flops, nbytes, *args, label=None, peak=None):
    if _prof is None:
        return _lib.call(name, *args)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _lib.call(name, *args)
    e1.record()
    if peak is None:
        fn = _PEAK_OF.get(name)
        peak = fn(args) if fn is not None else PEAK_FP32_MFMA
    _prof.setdefault(label or name, []).append((e0, e1, flops, nbytes, peak))


def _planes_peak(planes):
    return PEAK_BF16_MFMA / 6 if planes == 3 else PEAK_BF16_MFMA


# matrix-instruction ceiling per GEMM-family entry point, from its C argument list (positions as in include/wfae.h)
_PEAK_OF = {
    "wfae_conv1x1_fwd": lambda a: _gemm_peak(a[8], a[7]),                 # (.., NB, Cin, Cout, HW): M = Cout, K = Cin
    "wfae_conv1x1_fwd_stats": lambda a: _gemm_peak(a[8], a[7]),
    "wfae_conv1x1_fwd_bnact": lambda a: _gemm_peak(a[10], a[9]),
    "wfae_conv1x1_fwd_bf16": lambda a: PEAK_BF16_MFMA,
    "wfae_conv1x1_bwd_data": lambda a: _gemm_peak(a[4], a[5]),            # M = Cin, K = Cout
    "wfae_conv1x1_bwd_data_bf16": lambda a: PEAK_BF16_MFMA,
    # weight gradients: M = Cout, or Cin when the roles are swapped (Cin < Cout and Cin < 128); K = the pixels
    "wfae_conv1x1_bwd_weight": lambda a: _gemm_peak(a[4] if (a[4] < a[5] and a[4] < 128) else a[5], 1 << 20),
    "wfae_conv1x1_bwd_weight_bnact": lambda a: _gemm_peak(a[6] if (a[6] < a[7] and a[6] < 128) else a[7], 1 << 20),
    "wfae_conv1x1_bwd_weight_bf16": lambda a: PEAK_BF16_MFMA,
    "wfae_linear_fwd": lambda a: _gemm_peak(a[6], a[5]),
    "wfae_linear_bwd_data": lambda a: _gemm_peak(a[4], a[5]),
    "wfae_linear_bwd_data_splitk": lambda a: _gemm_peak(a[4], a[5]),
    "wfae_linear_bwd_weight": lambda a: _gemm_peak(a[5], a[3]),
    "wfae_linear_bwd_weight_splitk": lambda a: _gemm_peak(a[5], a[3]),
    "wfae_wino_gemm_down_split": lambda a: _planes_peak(a[4]),
    "wfae_wino_gemm_up_split": lambda a: _planes_peak(a[4]),
    "wfae_wino_gemm_wgrad_split": lambda a: _planes_peak(a[4]),
    "wfae_wino_gemm_down": lambda a: _default_peak(),
    "wfae_wino_gemm_up": lambda a: _default_peak(),
    "wfae_wino_gemm_wgrad": lambda a: _default_peak(),
    "wfae_split_gemm": lambda a: _planes_peak(a[1]),
    "wfae_conv4x4s2_down": lambda a: _default_peak(),
    "wfae_conv4x4s2_up": lambda a: _default_peak(),
    "wfae_conv4x4s2_wgrad": lambda a: _default_peak(),
    "wfae_conv4x4s1_fwd": lambda a: _gemm_peak(a[5], 16 * a[4]),
    "wfae_conv4x4s1_bwd_weight": lambda a: _gemm_peak(a[5], 1 << 20),
}


def workspace(min_bytes: int = 0):
    """Per-(device, stream) scratch buffer handed to kernels that need one."""
    dev = torch.cuda.current_device()
    key = (dev, _stream())
    need = max(_WS_DEFAULT, int(min_bytes))
    buf = _ws.get(key)
    if buf is None or buf.numel() < need:
        buf = torch.empty(need, dtype=torch.uint8, device=f"cuda:{dev}")
        _ws[key] = buf
    return buf


def _chk(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise _lib.WfaeError("wfae kernels need device tensors (the HIP path has no CPU fallback)")
        if t.dtype != torch.float32:
            raise _lib.WfaeError(f"wfae kernels are fp32, got {t.dtype}")
        if not t.is_contiguous():
            raise _lib.WfaeError("wfae kernels need contiguous NCHW tensors")


BF16 = torch.bfloat16


def _chka(*ts):
    """like _chk for ACTIVATION tensors, which may be bf16 in the bf16-storage mode; all of them must agree -> the
    entry-point suffix ("" or "_bf16") and the element size"""
    dt = None
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise _lib.WfaeError("wfae kernels need device tensors (the HIP path has no CPU fallback)")
        if t.dtype not in (torch.float32, BF16):
            raise _lib.WfaeError(f"wfae activation tensors are fp32 or bf16, got {t.dtype}")
        if dt is not None and t.dtype != dt:
            raise _lib.WfaeError(f"activation tensors of one call must share a storage type ({dt} vs {t.dtype})")
        dt = t.dtype
        if not t.is_contiguous():
            raise _lib.WfaeError("wfae kernels need contiguous NCHW tensors")
    return ("_bf16", 2) if dt == BF16 else ("", 4)


# Activation storage (include/wfae.h "bf16 ACTIVATION STORAGE"): the dtype in which the convolution stacks keep their
# activations and activation gradients in HBM.  fp32 always at 'highest' / 'high' precision; at 'medium' bf16 unless
# WFAE_BF16_STORAGE=0 (then only the matrix-core operands are rounded, as in round 2).
_ACT_DTYPE = torch.float32


def activation_dtype():
    return _ACT_DTYPE


def set_activation_storage(dtype):
    """torch.float32 | torch.bfloat16 (bf16 needs 'medium' matmul precision)"""
    global _ACT_DTYPE
    if dtype not in (torch.float32, BF16):
        raise ValueError("activation storage is torch.float32 or torch.bfloat16")
    if dtype == BF16 and _lib.load().wfae_get_matmul_precision() != 1:
        raise _lib.WfaeError("bf16 activation storage needs set_float32_matmul_precision('medium')")
    _ACT_DTYPE = dtype


def to_f32(x):
    """bf16-stored tensor -> fp32 (the boundaries of the bf16-storage mode)"""
    if x.dtype == torch.float32:
        return x
    _chka(x)
    y = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    _call("wfae_convert_bf16_to_f32", 0, 6 * x.numel(), _p(x), _p(y), x.numel(), _stream())
    return y


def to_bf16(x):
    if x.dtype == BF16:
        return x
    _chk(x)
    y = torch.empty(x.shape, dtype=BF16, device=x.device)
    _call("wfae_convert_f32_to_bf16", 0, 6 * x.numel(), _p(x), _p(y), x.numel(), _stream())
    return y


def to_dtype(x, dtype):
    return to_bf16(x) if dtype == BF16 else to_f32(x)


def _p(t):
    return None if t is None else t.data_ptr()


# ----------------------------------------------- 1x1 conv, register-direct (c1r)
# csrc/c1r.hip: the Bottleneck's four 1x1 products on fp32 tensors at the C <= 256 stages (activation HBM -> registers ->
# bf16 matrix core with exact three-plane operands, no LDS staging).  The wfae_conv1x1_* wrappers below route to it.
_C1R = True   # A/B through set_c1r()


def set_c1r(on):
    """A/B switch: the fp32 1x1 forward / data gradient of the C <= 256 stages on csrc/c1r.hip or on gemm.hip"""
    global _C1R
    _C1R = bool(on)


def c1r_supported(m, k, hw):
    return (_C1R and _SPLIT_GEMM and _lib.load().wfae_get_matmul_precision() == 0
            and bool(_lib.load().wfae_c1r_supported(int(m), int(k), int(hw))))


def _c1r_take(m, k, hw, dgrad):
    """route this product to c1r?  Every shape it serves (profiles/r04_kbench_c1r_variants.txt, r04_kbench_c1r_knobs.txt): at
    C = 256 gemm.hip's kernels are bound by the fp32 MFMA instruction and c1r wins all four products (0.346 -> 0.31, 0.445 ->
    0.30, 0.685 -> 0.49, 0.500 -> 0.40 - 0.45 ms); at C = 128 both are HBM-bound: c1r wins the forward launches, which carry a
    BatchNorm + GELU prologue and / or a residual + BatchNorm-sum epilogue in the step (0.72 -> 0.65, 1.14 -> 1.02 ms), and
    ties the two plain data gradients (0.54 - 0.59 vs 0.57 - 0.59 ms) — one kernel family per stage, and every form of a
    shape on the same kernel keeps fused and unfused forms bit-identical.  The M-sliced shapes of the C >= 512 stages serve the
    widening products in both directions (profiles/r04_kbench_c1r_sliced.txt)."""
    return c1r_supported(m, k, hw)


def _c1r(w, transposed, x, st, res, stats, label):
    """y = A f(x) (+ res): A = w (Cout, Cin) or, transposed, w^T (the data gradient); -> y | (y, StatRows)"""
    import ctypes
    nb, k, h, wd = x.shape
    cout, cin = w.shape[0], w.shape[1]
    m, sm, sk = (cin, 1, cin) if transposed else (cout, cin, 1)
    y = torch.empty((nb, m, h, wd), dtype=torch.float32, device=x.device)
    fl = 2 * nb * h * wd * k * m
    by = 4 * (nb * h * wd * (k + m) + k * m) + (0 if res is None else 4 * nb * h * wd * m)
    ps, ph = (None, None) if st is None else (_p(st.scale), _p(st.shift))
    if not stats:
        _call("wfae_c1r_fwd", fl, by, _p(w), sm, sk, _p(x), ps, ph, _p(res), _p(y), nb, k, m, h * wd, None, 0, None, _stream(),
              label=label, peak=PEAK_BF16_MFMA / 6)
        return y
    rows_n = int(_lib.load().wfae_c1r_stat_rows(m, k, nb, h * wd))
    part = torch.empty(2 * rows_n * m, dtype=torch.float64, device=x.device)
    rows = ctypes.c_int(0)
    _call("wfae_c1r_fwd", fl, by, _p(w), sm, sk, _p(x), ps, ph, _p(res), _p(y), nb, k, m, h * wd, part.data_ptr(), part.numel(),
          ctypes.cast(ctypes.pointer(rows), ctypes.c_void_p), _stream(), label=label, peak=PEAK_BF16_MFMA / 6)
    return y, StatRows(part[:2 * rows.value * m], rows.value)     # (the capacity is sized for the widest user of the shape)


_C1R_BNRED = True   # A/B through set_c1r_bnred()


def set_c1r_bnred(on):
    """A/B switch: the reductions of the first BatchNorm's backward pass in the epilogue of the c1r data gradient (C <= 256)"""
    global _C1R_BNRED
    _C1R_BNRED = bool(on)


def c1r_bnred_supported(m, k, hw):
    return (_C1R and _C1R_BNRED and _SPLIT_GEMM and _lib.load().wfae_get_matmul_precision() == 0
            and bool(_lib.load().wfae_c1r_bnred_supported(int(m), int(k), int(hw))))


def c1r_bnred(w, dt, x, st, store=True):
    """da = conv1x1_bwd_data(dt, w) on c1r with phase 1 of bn_act_bwd(da, x, ...) taken in the epilogue -> (da, StatRows of
    (sum dU, sum dU xhat)); finish with bn_act_bwd_from_rows, then bn_act_bwd_dx — or, with store=False (da is None: the sums
    alone), with c1r_bndx, which rebuilds da in its own registers"""
    import ctypes
    _chk(w, dt, x)
    nb, k, h, wd = dt.shape
    m = w.shape[1]
    da = torch.empty((nb, m, h, wd), dtype=torch.float32, device=dt.device) if store else None
    rows_n = int(_lib.load().wfae_c1r_stat_rows(m, k, nb, h * wd))
    part = torch.empty(2 * rows_n * m, dtype=torch.float64, device=dt.device)
    rows = ctypes.c_int(0)
    n = nb * h * wd
    _call("wfae_c1r_bnred", 2 * n * k * m, 4 * (n * (k + (2 if store else 1) * m) + k * m), _p(w), 1, m, _p(dt), _p(x), _p(st.scale), _p(st.shift),
          _p(st.mean), _p(st.invstd), _p(da), nb, k, m, h * wd, part.data_ptr(), part.numel(),
          ctypes.cast(ctypes.pointer(rows), ctypes.c_void_p), _stream(), label="wfae_conv1x1_bwd_data", peak=PEAK_BF16_MFMA / 6)
    return da, StatRows(part[:2 * rows.value * m], rows.value)


def bn_act_bwd_from_rows(sr, c, dgamma, dbeta, accumulate=False):
    """phase 1 of bn_act_bwd from the partial rows of c1r_bnred: dgamma / dbeta + the coefficients at the head of this stream's
    workspace, where bn_act_bwd_dx (phase 2) reads them (no other workspace user in between)"""
    _chk(dgamma, dbeta)
    ws = workspace()
    _call("wfae_bn_act_bwd_from_rows", 0, 8 * sr.part.numel(), sr.part.data_ptr(), sr.rows, c, _p(dgamma), _p(dbeta), int(accumulate),
          ws.data_ptr(), ws.numel(), _stream(), label="wfae_bn_act_bwd[reduce]")
    return ws


def bn_act_bwd_dx(dy, x, gamma, st, res=None, act=1, training=True):
    """phase 2 of bn_act_bwd alone (after bn_act_bwd_from_rows on the same stream; fp32 storage)"""
    _chk(dy, x, gamma, res)
    nb, c, h, wd = x.shape
    dx = torch.empty_like(x)
    ws = workspace()
    _call("wfae_bn_act_bwd", 0, 4 * x.numel() * (4 if res is not None else 3), _p(dy), _p(x), _p(gamma), _p(st.scale), _p(st.shift),
          _p(st.mean), _p(st.invstd), _p(res), _p(dx), None, None, nb, c, h * wd, act, int(training), 0, 2, ws.data_ptr(),
          ws.numel(), _stream(), label="wfae_bn_act_bwd[dx]")
    return dx


_C1R_BNDX = True   # A/B through set_c1r_bndx()


def set_c1r_bndx(on):
    """A/B switch: the second pass of the first BatchNorm's backward in the epilogue of a recomputed data gradient (c1r_bndx)
    or as bn_act_bwd_dx over the stored one"""
    global _C1R_BNDX
    _C1R_BNDX = bool(on)


def c1r_bndx_on():
    return _C1R_BNDX


def c1r_bndx(w, dt, x, gamma, st, res=None, training=True):
    """dx of  conv1x1(gelu(bn(x)), w)  given dt, second pass: da = conv1x1_bwd_data(dt, w) rebuilt in registers and
    bn_act_bwd_dx(da, x, ...) + res applied in the epilogue (after c1r_bnred(store=False) + bn_act_bwd_from_rows on this stream)"""
    _chk(w, dt, x, gamma, res)
    nb, k, h, wd = dt.shape
    m = w.shape[1]
    dx = torch.empty((nb, m, h, wd), dtype=torch.float32, device=dt.device)
    ws = workspace()
    n = nb * h * wd
    _call("wfae_c1r_bndx", 2 * n * k * m, 4 * (n * (k + (3 if res is not None else 2) * m) + k * m), _p(w), 1, m, _p(dt), _p(x), _p(gamma),
          _p(st.scale), _p(st.shift), _p(st.mean), _p(st.invstd), ws.data_ptr(), _p(res), _p(dx), nb, k, m, h * wd, int(training),
          _stream(), label="wfae_c1r_bndx", peak=PEAK_BF16_MFMA / 6)
    return dx


# csrc/c1rb.hip: the same register-direct product on bf16-stored tensors ('medium'): every Bottleneck stage with HW % 128 == 0
_C1RB = True   # A/B through set_c1rb()


def set_c1rb(on):
    """A/B switch: the bf16 1x1 forward / data gradient on csrc/c1rb.hip (register-direct) or on csrc/c1b.hip (LDS-tiled)"""
    global _C1RB
    _C1RB = bool(on)


def c1rb_supported(m, k, hw):
    return _C1RB and _lib.load().wfae_get_matmul_precision() == 1 and bool(_lib.load().wfae_c1rb_supported(int(m), int(k), int(hw)))


def c1rb_take(m, k, hw, pro):
    """route this bf16 product to c1rb?  Measured (profiles/r04_kbench_c1rb_vs_c1b.txt): it wins or ties every form except
    the narrowing products of C >= 256 with the BatchNorm + GELU prologue, which stay on c1b: the M-sliced ones (C >= 512)
    evaluate the activation of the whole operand again in every slice (0.163 -> 0.462 ms), and the prologue forms run one wave
    per SIMD (csrc/c1rb.hip, wfae_c1rb_fwd: with two, repeat launches differed), where C = 256 takes 0.327 ms against c1b's
    0.284 (C = 128: 0.477 against 0.536, kept)."""
    if pro and m < k and k >= 256:
        return False
    return c1rb_supported(m, k, hw)


def c1rb_fwd(w, transposed, x, st=None, res=None, stats=False, label="wfae_c1b_fwd"):
    """y (bf16) = A f(x) (+ res) on bf16-stored activations: A = w (Cout, Cin) or, transposed, w^T (the data gradient); the
    fp32 weight itself is passed (the kernel rounds it into its LDS image) -> y | (y, StatRows)"""
    import ctypes
    sfx, _ = _chka(x, res)
    _chk(w)
    if not sfx:
        raise _lib.WfaeError("c1rb_fwd: bf16-stored activations")
    nb, k, h, wd = x.shape
    cout, cin = w.shape[0], w.shape[1]
    m, sm, sk = (cin, 1, cin) if transposed else (cout, cin, 1)
    y = torch.empty((nb, m, h, wd), dtype=BF16, device=x.device)
    fl = 2 * nb * h * wd * k * m
    by = 2 * nb * h * wd * (k + m) + 4 * k * m + (0 if res is None else 2 * nb * h * wd * m)
    ps, ph = (None, None) if st is None else (_p(st.scale), _p(st.shift))
    if not stats:
        _call("wfae_c1rb_fwd", fl, by, _p(w), sm, sk, _p(x), ps, ph, _p(res), _p(y), nb, k, m, h * wd, None, 0, None, _stream(),
              label=label, peak=PEAK_BF16_MFMA)
        return y
    rows_n = int(_lib.load().wfae_c1rb_stat_rows(m, k, nb, h * wd))
    part = torch.empty(2 * rows_n * m, dtype=torch.float64, device=x.device)
    rows = ctypes.c_int(0)
    _call("wfae_c1rb_fwd", fl, by, _p(w), sm, sk, _p(x), ps, ph, _p(res), _p(y), nb, k, m, h * wd, part.data_ptr(), part.numel(),
          ctypes.cast(ctypes.pointer(rows), ctypes.c_void_p), _stream(), label=label, peak=PEAK_BF16_MFMA)
    return y, StatRows(part[:2 * rows.value * m], rows.value)


# ----------------------------------------------------------------- 1x1 conv
def _conv1x1_fwd_bf16(x, st, w, bias, res, stats):
    """the merged bf16-storage entry point: optional BatchNorm + GELU prologue (st), optional BatchNorm sums (stats)"""
    import ctypes
    _chka(x, res)
    _chk(w, bias)
    nb, cin, h, wd = x.shape
    cout = w.shape[0]
    y = torch.empty((nb, cout, h, wd), dtype=BF16, device=x.device)
    fl = 2 * nb * h * wd * cin * cout
    by = 2 * nb * h * wd * (cin + cout) + 4 * cin * cout + (0 if res is None else 2 * nb * h * wd * cout)
    ps, ph = (None, None) if st is None else (_p(st.scale), _p(st.shift))
    label = "wfae_conv1x1_fwd" if st is None else "wfae_conv1x1_fwd_bnact"
    if not stats:
        _call("wfae_conv1x1_fwd_bf16", fl, by, _p(x), ps, ph, _p(w), _p(bias), _p(res), cout * h * wd, _p(y), nb, cin, cout, h * wd,
              None, 0, None, _stream(), label=label)
        return y
    part, cap = _stat_rows_buffer(nb, h * wd, cout, x.device)
    rows = ctypes.c_int(0)
    _call("wfae_conv1x1_fwd_bf16", fl, by, _p(x), ps, ph, _p(w), _p(bias), _p(res), cout * h * wd, _p(y), nb, cin, cout, h * wd,
          part.data_ptr(), cap, ctypes.cast(ctypes.pointer(rows), ctypes.c_void_p), _stream(), label=label)
    return y, (StatRows(part, rows.value) if rows.value > 0 else None)


def conv1x1_fwd(x, w, bias=None, res=None, res_broadcast=False):
    if x.dtype == BF16:
        if res_broadcast:
            raise _lib.WfaeError("conv1x1_fwd: the broadcast residual (pos_emb) belongs to an fp32 layer")
        return _conv1x1_fwd_bf16(x, None, w, bias, res, False)
    _chk(x, w, bias, res)
    nb, cin, h, wd = x.shape
    cout = w.shape[0]
    if bias is None and not res_broadcast and (res is None or cout > cin) and _c1r_take(cout, cin, h * wd, False):
        return _c1r(w, False, x, None, res, False, "wfae_conv1x1_fwd")
    y = torch.empty((nb, cout, h, wd), dtype=x.dtype, device=x.device)
    stride = 0 if res_broadcast else cout * h * wd
    _call("wfae_conv1x1_fwd", 2 * nb * h * wd * cin * cout, 4 * (nb * h * wd * (cin + cout) + cin * cout) + (0 if res is None else 4 * nb * h * wd * cout), _p(x), _p(w), _p(bias), _p(res), stride, _p(y), nb, cin, cout, h * wd, _stream())
    return y


class StatRows:
    """partial BatchNorm sums written by a GEMM epilogue: `part` (float64) = sum[rows][C] ++ sumsq[rows][C]"""
    __slots__ = ("part", "rows")

    def __init__(self, part, rows):
        self.part, self.rows = part, rows


class StatParts:
    """fp64 partial BatchNorm sums left by a producer kernel: `part` = [splits][C][2] doubles (sum, sum of squares)"""
    __slots__ = ("part", "splits")

    def __init__(self, part, splits):
        self.part, self.splits = part, splits


def _stat_rows_buffer(nb, hw, cout, device):
    cap = 4 * ((nb * hw + 127) // 128) * cout
    return torch.empty(cap, dtype=torch.float64, device=device), cap


def conv1x1_fwd_stats(x, w, bias=None, res=None):
    """conv1x1_fwd whose epilogue also reduces the per-channel sum / sum of squares of y (for the BatchNorm that
    follows).  -> (y, StatRows | None); None = this shape is not served, run bn_stats_train on y."""
    import ctypes
    if x.dtype == BF16:
        return _conv1x1_fwd_bf16(x, None, w, bias, res, True)
    _chk(x, w, bias, res)
    nb, cin, h, wd = x.shape
    cout = w.shape[0]
    if bias is None and (res is None or cout > cin) and _c1r_take(cout, cin, h * wd, False):
        return _c1r(w, False, x, None, res, True, "wfae_conv1x1_fwd")
    y = torch.empty((nb, cout, h, wd), dtype=x.dtype, device=x.device)
    part, cap = _stat_rows_buffer(nb, h * wd, cout, x.device)
    rows = ctypes.c_int(0)
    _call("wfae_conv1x1_fwd_stats", 2 * nb * h * wd * cin * cout,
          4 * (nb * h * wd * (cin + cout) + cin * cout) + (0 if res is None else 4 * nb * h * wd * cout),
          _p(x), _p(w), _p(bias), _p(res), cout * h * wd, _p(y), nb, cin, cout, h * wd, part.data_ptr(), cap,
          ctypes.cast(ctypes.pointer(rows), ctypes.c_void_p), _stream(), label="wfae_conv1x1_fwd")
    return y, (StatRows(part, rows.value) if rows.value > 0 else None)


def conv1x1_bnact_supported(x, cout):
    """geometries wfae_conv1x1_fwd_bnact / wfae_conv1x1_bwd_weight_bnact serve (the library re-checks and refuses the rest)"""
    nb, cin, h, wd = x.shape
    swapped = cin < cout and cin < 128          # the weight gradient then computes dW^T (N = Cout columns)
    return (h * wd) % 4 == 0 and h * wd >= 16 and cin % 4 == 0 and (cout % 4 == 0 or not swapped)


def conv1x1_fwd_bnact(x, st, w, bias=None, res=None, stats=False):
    """conv1x1_fwd(bn_act_fwd(x, st, GELU), w): the activated tensor is rebuilt in the GEMM's operand loader and never
    written (reference chain BN -> GELU -> Conv2d 1x1, pipeline/models/ae_64x8x8_lin.py:14-15).  stats=True: the
    epilogue also reduces the BatchNorm sums of y -> (y, StatRows | None)"""
    import ctypes
    if x.dtype == BF16:
        return _conv1x1_fwd_bf16(x, st, w, bias, res, stats)
    _chk(x, w, bias, res)
    nb, cin, h, wd = x.shape
    cout = w.shape[0]
    if bias is None and (res is None or cout > cin) and _c1r_take(cout, cin, h * wd, False):
        return _c1r(w, False, x, st, res, stats, "wfae_conv1x1_fwd_bnact")
    y = torch.empty((nb, cout, h, wd), dtype=x.dtype, device=x.device)
    fl = 2 * nb * h * wd * cin * cout
    by = 4 * (nb * h * wd * (cin + cout) + cin * cout) + (0 if res is None else 4 * nb * h * wd * cout)
    if not stats:
        _call("wfae_conv1x1_fwd_bnact", fl, by, _p(x), _p(st.scale), _p(st.shift), _p(w), _p(bias), _p(res), cout * h * wd, _p(y),
              nb, cin, cout, h * wd, None, 0, None, _stream())
        return y
    part, cap = _stat_rows_buffer(nb, h * wd, cout, x.device)
    rows = ctypes.c_int(0)
    _call("wfae_conv1x1_fwd_bnact", fl, by, _p(x), _p(st.scale), _p(st.shift), _p(w), _p(bias), _p(res), cout * h * wd, _p(y),
          nb, cin, cout, h * wd, part.data_ptr(), cap, ctypes.cast(ctypes.pointer(rows), ctypes.c_void_p), _stream())
    return y, (StatRows(part, rows.value) if rows.value > 0 else None)


# weight gradients on csrc/c1w.hip (both operands as K-contiguous rows, no LDS transpose): fp32 tensors at fp32 precision with
# the split GEMMs on, bf16-stored tensors at 'medium'
_C1W = True   # A/B through set_c1w()


def set_c1w(on):
    global _C1W
    _C1W = bool(on)


def _c1w_route(dy, x, sfx):
    nb, cout, h, wd = dy.shape
    if not (_C1W and _lib.load().wfae_c1w_supported(x.shape[1], cout, h * wd)):
        return False
    prec = _lib.load().wfae_get_matmul_precision()
    return prec == 1 if sfx else (prec == 0 and _SPLIT_GEMM)


def _c1w(dy, x, st, dw, accumulate, sfx, es, label):
    nb, cout, h, wd = dy.shape
    cin = x.shape[1]
    ws = workspace()
    ps, ph = (None, None) if st is None else (_p(st.scale), _p(st.shift))
    _call("wfae_c1w_bwd_weight" + sfx, 2 * nb * h * wd * cin * cout, es * nb * h * wd * (cin + cout) + 4 * cin * cout, _p(dy), _p(x),
          ps, ph, _p(dw), nb, cin, cout, h * wd, int(accumulate), ws.data_ptr(), ws.numel(), _stream(), label=label,
          peak=PEAK_BF16_MFMA / (1 if sfx else 6))
    return dw


def conv1x1_bwd_weight_bnact(dy, x, st, dw, accumulate=False):
    """conv1x1_bwd_weight(dy, bn_act_fwd(x, st, GELU), dw) without the activated tensor in HBM"""
    sfx, es = _chka(dy, x)
    _chk(dw)
    if _c1w_route(dy, x, sfx):
        return _c1w(dy, x, st, dw, accumulate, sfx, es, "wfae_conv1x1_bwd_weight_bnact")
    nb, cout, h, wd = dy.shape
    cin = x.shape[1]
    ws = workspace()
    if sfx:
        _call("wfae_conv1x1_bwd_weight_bf16", 2 * nb * h * wd * cin * cout, 2 * nb * h * wd * (cin + cout) + 4 * cin * cout, _p(dy),
              _p(x), _p(st.scale), _p(st.shift), _p(dw), nb, cin, cout, h * wd, int(accumulate), ws.data_ptr(), ws.numel(),
              _stream(), label="wfae_conv1x1_bwd_weight_bnact")
        return dw
    _call("wfae_conv1x1_bwd_weight_bnact", 2 * nb * h * wd * cin * cout, 4 * (nb * h * wd * (cin + cout) + cin * cout),
          _p(dy), _p(x), _p(st.scale), _p(st.shift), _p(dw), nb, cin, cout, h * wd, int(accumulate), ws.data_ptr(),
          ws.numel(), _stream())
    return dw


def conv1x1_bwd_data(dy, w):
    sfx, es = _chka(dy)
    _chk(w)
    nb, cout, h, wd = dy.shape
    cin = w.shape[1]
    if not sfx and _c1r_take(cin, cout, h * wd, True):
        return _c1r(w, True, dy, None, None, False, "wfae_conv1x1_bwd_data")
    dx = torch.empty((nb, cin, h, wd), dtype=dy.dtype, device=dy.device)
    _call("wfae_conv1x1_bwd_data" + sfx, 2 * nb * h * wd * cin * cout, es * nb * h * wd * (cin + cout) + 4 * cin * cout, _p(dy), _p(w),
          _p(dx), nb, cin, cout, h * wd, _stream(), label="wfae_conv1x1_bwd_data")
    return dx


def conv1x1_bwd_weight(dy, x, dw, accumulate=False):
    sfx, es = _chka(dy, x)
    _chk(dw)
    if _c1w_route(dy, x, sfx):
        return _c1w(dy, x, None, dw, accumulate, sfx, es, "wfae_conv1x1_bwd_weight")
    nb, cout, h, wd = dy.shape
    cin = x.shape[1]
    ws = workspace()
    if sfx:
        _call("wfae_conv1x1_bwd_weight_bf16", 2 * nb * h * wd * cin * cout, 2 * nb * h * wd * (cin + cout) + 4 * cin * cout, _p(dy),
              _p(x), None, None, _p(dw), nb, cin, cout, h * wd, int(accumulate), ws.data_ptr(), ws.numel(), _stream(),
              label="wfae_conv1x1_bwd_weight")
        return dw
    _call("wfae_conv1x1_bwd_weight", 2 * nb * h * wd * cin * cout, 4 * (nb * h * wd * (cin + cout) + cin * cout), _p(dy), _p(x), _p(dw), nb, cin, cout, h * wd, int(accumulate),
              ws.data_ptr(), ws.numel(), _stream())
    return dw


# ---- bf16 storage: csrc/c1b.hip (no format change between HBM and the matrix core)
_C1B = True   # A/B through set_c1b()


def set_c1b(on):
    """A/B switch: bf16 1x1 forward / data gradient on csrc/c1b.hip or on the element-typed generic GEMM kernel"""
    global _C1B
    _C1B = bool(on)


def c1b_supported(m, k, hw):
    return _C1B and _lib.load().wfae_get_matmul_precision() == 1 and bool(_lib.load().wfae_c1b_supported(int(m), int(k), int(hw)))


def c1b_weights(w):
    """w (Cout, Cin[,1,1]) fp32 -> (Wb (Cout, Cin), Wtb (Cin, Cout)) bf16"""
    _chk(w)
    cout, cin = w.shape[0], w.shape[1]
    Wb = torch.empty((cout, cin), dtype=BF16, device=w.device)
    Wtb = torch.empty((cin, cout), dtype=BF16, device=w.device)
    _call("wfae_c1b_weights", 0, 8 * w.numel(), _p(w), _p(Wb), _p(Wtb), cout, cin, _stream())
    return Wb, Wtb


def c1b_fwd(Wb, x, st=None, res=None, stats=False, label="wfae_c1b_fwd"):
    """y (bf16) = W f(x) (+ res) on bf16-stored activations; Wb (M, K) from c1b_weights; st: BnStats folded into the
    operand loader (BatchNorm + GELU in front of the convolution); stats=True: -> (y, StatRows of y)"""
    import ctypes
    sfx, _ = _chka(x, res)
    if not sfx or Wb.dtype != BF16 or not Wb.is_contiguous() or Wb.shape[1] != x.shape[1]:
        raise _lib.WfaeError("c1b_fwd: bf16 activations and the contiguous (M, K) bf16 weight plane of c1b_weights")
    nb, k, h, wd = x.shape
    m = Wb.shape[0]
    y = torch.empty((nb, m, h, wd), dtype=BF16, device=x.device)
    fl = 2 * nb * h * wd * k * m
    by = 2 * nb * h * wd * (k + m) + 2 * k * m + (0 if res is None else 2 * nb * h * wd * m)
    ps, ph = (None, None) if st is None else (_p(st.scale), _p(st.shift))
    if not stats:
        _call("wfae_c1b_fwd", fl, by, _p(Wb), _p(x), ps, ph, _p(res), _p(y), nb, k, m, h * wd, None, 0, None, _stream(), label=label,
              peak=PEAK_BF16_MFMA)
        return y
    rows_n = int(_lib.load().wfae_c1b_stat_rows(m, k, nb, h * wd))
    part = torch.empty(2 * rows_n * m, dtype=torch.float64, device=x.device)
    rows = ctypes.c_int(0)
    _call("wfae_c1b_fwd", fl, by, _p(Wb), _p(x), ps, ph, _p(res), _p(y), nb, k, m, h * wd, part.data_ptr(), part.numel(),
          ctypes.cast(ctypes.pointer(rows), ctypes.c_void_p), _stream(), label=label, peak=PEAK_BF16_MFMA)
    return y, StatRows(part, rows.value)


# ------------------------------------------------------------------- linear
def linear_fwd(x, w, bias=None):
    _chk(x, w, bias)
    b, inf = x.shape
    out = w.shape[0]
    y = torch.empty((b, out), dtype=x.dtype, device=x.device)
    ws = workspace()
    _call("wfae_linear_fwd", 2 * b * inf * out, 4 * (inf * out + b * (inf + out)), _p(x), _p(w), _p(bias), _p(y), b, inf, out, ws.data_ptr(), ws.numel(), _stream())
    return y


def linear_bwd_data(dy, w):
    _chk(dy, w)
    b, out = dy.shape
    inf = w.shape[1]
    dx = torch.empty((b, inf), dtype=dy.dtype, device=dy.device)
    tiles = ((b + 127) // 128) * ((inf + 127) // 128)
    if tiles < 256 and out >= 512:    # few output tiles, long reduction: split it
        ws = workspace(b * inf * 4 * 2)
        _call("wfae_linear_bwd_data_splitk", 2 * b * inf * out, 4 * (inf * out + b * (inf + out)), _p(dy), _p(w), _p(dx), b, inf,
              out, ws.data_ptr(), ws.numel(), _stream())
        return dx
    _call("wfae_linear_bwd_data", 2 * b * inf * out, 4 * (inf * out + b * (inf + out)), _p(dy), _p(w), _p(dx), b, inf, out, _stream())
    return dx


def linear_bwd_weight(dy, x, dw, accumulate=False):
    _chk(dy, x, dw)
    b, out = dy.shape
    inf = x.shape[1]
    tiles = ((out + 127) // 128) * ((inf + 127) // 128)
    if tiles < 256 and b >= 256:      # few output tiles, long batch dimension: split it (transformer layers)
        ws = workspace(out * inf * 4 * 2)
        _call("wfae_linear_bwd_weight_splitk", 2 * b * inf * out, 4 * (inf * out + b * (inf + out)), _p(dy), _p(x), _p(dw), b, inf,
              out, int(accumulate), ws.data_ptr(), ws.numel(), _stream())
        return dw
    _call("wfae_linear_bwd_weight", 2 * b * inf * out, 4 * (inf * out + b * (inf + out)), _p(dy), _p(x), _p(dw), b, inf, out, int(accumulate), _stream())
    return dw


# ------------------------------------------------------ 4x4 stride-2 family
# Winograd forms of the 4x4 stride-2 operations (csrc/wino.hip, include/wfae.h): F(2x2,2x2) (9/16 of the
# multiplies) or F(4x4,2x2) (25/64), paid for with streaming transforms of the operands.
# WFAE_WINO = 0: never; f22 / f42: that variant whenever the geometry allows; unset / auto: F(4x4,2x2) when Hlo, Wlo
# are multiples of 4, else F(2x2,2x2), for layers with both channel counts >= 16 (measured, tools/kbench.py: every
# GEMM-class layer of the model gains).
_WINO_MODE = os.environ.get("WFAE_WINO", "auto")
_plans = {}


def set_winograd(mode):
    """'auto' | 'f22' | 'f42' | True (= auto without the channel threshold) | False — overrides WFAE_WINO"""
    global _WINO_MODE
    _WINO_MODE = mode if isinstance(mode, str) else ("1" if mode else "0")


_PRECISIONS = {"highest": 0, "high": 0, "medium": 1}


def set_float32_matmul_precision(precision):
    """Counterpart of torch.set_float32_matmul_precision, which the reference calls once at start-up
    (experiments/ae_v2/train.py:270, 'high' = TF32 on its CUDA machines).  'highest' (default) and 'high': fp32
    MFMA (gfx950 has no TF32/xf32 matrix instruction; fp32 is the next precision up).  'medium': every GEMM-family
    kernel rounds its operands to bf16 on the way into the matrix core and accumulates in fp32, tensors stay fp32
    (BASELINE config 5).  In 'medium' the automatic Winograd choice is F(2x2,2x2): the F(4x4,2x2) transforms
    amplify operand rounding ~10x, which fp32 absorbs and bf16 does not."""
    global _ACT_DTYPE
    if precision not in _PRECISIONS:
        raise ValueError(f"precision must be one of {sorted(_PRECISIONS)}, got {precision!r}")
    _lib.call("wfae_set_matmul_precision", _PRECISIONS[precision])
    _ACT_DTYPE = BF16 if (_PRECISIONS[precision] == 1 and os.environ.get("WFAE_BF16_STORAGE", "1") != "0") else torch.float32


def get_float32_matmul_precision():
    return "medium" if _lib.load().wfae_get_matmul_precision() == 1 else "highest"


class WinoPlan:
    __slots__ = ("variant", "nb", "chi", "clo", "hlo", "wlo", "T", "nU", "nV", "nM", "split", "planes")

    @property
    def dims(self):
        return (self.nb, self.chi, self.clo, self.hlo, self.wlo)

    @property
    def gemm_flops(self):
        per = 12.5 if self.variant else 18.0
        return per * self.nb * self.hlo * self.wlo * self.clo * self.chi


def _query_plan(variant, key):
    import ctypes
    out = (ctypes.c_int64 * 4)()
    rc = _lib.load().wfae_wino_sizes(variant, *key, ctypes.cast(out, ctypes.c_void_p))
    if rc != 0:
        return None
    pl = WinoPlan()
    pl.variant = variant
    pl.nb, pl.chi, pl.clo, pl.hlo, pl.wlo = key
    pl.T, pl.nU, pl.nV, pl.nM = (int(v) for v in out)
    pl.split = False
    pl.planes = 3
    return pl


# Winograd-domain GEMMs on the bf16 matrix pipe with fp32-exact three-plane operands (include/wfae.h, "split"): the
# default at fp32 precision wherever the geometry allows; WFAE_SPLIT_GEMM=0 keeps v_mfma_f32_32x32x2_f32
_SPLIT_GEMM = os.environ.get("WFAE_SPLIT_GEMM", "1") != "0"


def set_split_gemm(on):
    """switch the split-operand GEMMs (Winograd-domain products and the in-register split of the generic GEMM kernel)
    on / off (plans are cached per setting)"""
    global _SPLIT_GEMM
    _SPLIT_GEMM = bool(on)
    _lib.call("wfae_set_split_gemm", int(_SPLIT_GEMM))


def split_gemm_enabled():
    return _SPLIT_GEMM


def wino_plan(nb, chi, clo, hlo, wlo):
    """WinoPlan when a Winograd form should run for this layer geometry, else None"""
    mode = _WINO_MODE
    if mode == "0":
        return None
    if mode == "auto" and min(chi, clo) < 16:
        return None
    if mode not in ("f22", "f42") and _lib.load().wfae_get_matmul_precision() == 1:
        mode = os.environ.get("WFAE_WINO_BF16", "f22")
        if mode == "0":
            return None
    key = (nb, chi, clo, hlo, wlo)
    # split operands: three exact planes at fp32 precision, the bf16 h plane alone at 'medium' precision
    prec = _lib.load().wfae_get_matmul_precision()
    split = _SPLIT_GEMM
    ck = (mode if mode in ("f22", "f42") else "a", split, prec, key)
    pl = _plans.get(ck, False)
    if pl is False:
        if mode == "f22":
            pl = _query_plan(0, key)
        elif mode == "f42":
            pl = _query_plan(1, key)
        else:
            pl = _query_plan(1, key) or _query_plan(0, key)
        if pl is not None and split:
            pl.split = bool(_lib.load().wfae_wino_split_supported(pl.variant, *key))
            pl.planes = 3 if prec == 0 else 1
        _plans[ck] = pl
    return pl


def _buf(n, like):
    return torch.empty(n, dtype=torch.float32, device=like.device)


def _buf3(n, like, planes=3):
    """a split operand: `planes` planes of bf16 bit patterns, n elements each"""
    return torch.empty(planes * n, dtype=torch.int16, device=like.device)


def split_bf16x3(x, planes=3):
    """fp32 tensor -> its three bf16 planes (int16 bit patterns, shape (3,) + x.shape): x == h + m + l exactly;
    planes=1: the h plane alone (x rounded to bf16)"""
    _chk(x)
    out = torch.empty((planes,) + tuple(x.shape), dtype=torch.int16, device=x.device)
    _call("wfae_split_bf16x3", 0, (4 + 2 * planes) * x.numel(), _p(x), out.data_ptr(), x.numel(), planes, _stream())
    return out


def split_gemm(a3, b3, b_kind=0):
    """C[y] = A[y] B[y] from split operands: a3 (P, Y, M, K); b3 (P, Y, K, N) for b_kind 0, (P, Y, N, K) for b_kind 1;
    P = 3 planes (exact fp32 product) or 1 (bf16-rounded operands)"""
    planes, y, m, k = a3.shape
    n = b3.shape[3] if b_kind == 0 else b3.shape[2]
    if a3.dtype != torch.int16 or b3.dtype != torch.int16 or not (a3.is_contiguous() and b3.is_contiguous()):
        raise _lib.WfaeError("split_gemm: contiguous int16 plane tensors from split_bf16x3")
    c = torch.empty((y, m, n), dtype=torch.float32, device=a3.device)
    if b3.shape[0] != planes:
        raise _lib.WfaeError("split_gemm: operands with different plane counts")
    _call("wfae_split_gemm", 2 * y * m * n * k, 2 * planes * y * k * (m + n) + 4 * y * m * n, b_kind, planes, a3.data_ptr(),
          b3.data_ptr(), _p(c), m, n, k, y, _stream())
    return c


def wino_weights(w, pl):
    _chk(w)
    if pl.split:
        U = _buf3(2 * pl.nU, w, pl.planes)   # U3 then Ut3
        _call("wfae_wino_weights_split", 0, 4 * w.numel() + 4 * pl.planes * pl.nU, pl.variant, _p(w), U.data_ptr(),
              U.data_ptr() + 2 * pl.planes * pl.nU, pl.planes, pl.chi, pl.clo, _stream(), label="wfae_wino_weights")
        return U
    U = _buf(pl.nU, w)
    _call("wfae_wino_weights", 0, 4 * (w.numel() + pl.nU), pl.variant, _p(w), _p(U), pl.chi, pl.clo, _stream())
    return U


def wino_in(hi, pl):
    """hi-side tensor (N,Chi,2Hlo,2Wlo) -> V[xi][4Chi][T]"""
    if not pl.split:
        hi = to_f32(hi)      # the fp32-operand transforms read fp32 tensors
    sfx, es = _chka(hi)
    if pl.split:
        V = _buf3(pl.nV, hi, pl.planes)
        _call("wfae_wino_in_split" + sfx, 0, es * hi.numel() + 2 * pl.planes * pl.nV, pl.variant, _p(hi), V.data_ptr(), pl.planes,
              pl.nb, pl.chi, pl.hlo, pl.wlo, _stream(), label="wfae_wino_in")
        return V
    V = _buf(pl.nV, hi)
    _call("wfae_wino_in", 0, 4 * (hi.numel() + pl.nV), pl.variant, _p(hi), _p(V), pl.nb, pl.chi, pl.hlo, pl.wlo, _stream())
    return V


def wino_out_t(lo, pl):
    """lo-side tensor (N,Clo,Hlo,Wlo) -> Mt[xi][Clo][T]"""
    if not pl.split:
        lo = to_f32(lo)
    sfx, es = _chka(lo)
    if pl.split:
        Mt = _buf3(pl.nM, lo, pl.planes)
        _call("wfae_wino_out_t_split" + sfx, 0, es * lo.numel() + 2 * pl.planes * pl.nM, pl.variant, _p(lo), Mt.data_ptr(), pl.planes,
              pl.nb, pl.clo, pl.hlo, pl.wlo, _stream(), label="wfae_wino_out_t")
        return Mt
    Mt = _buf(pl.nM, lo)
    _call("wfae_wino_out_t", 0, 4 * (lo.numel() + pl.nM), pl.variant, _p(lo), _p(Mt), pl.nb, pl.clo, pl.hlo, pl.wlo, _stream())
    return Mt


# tiles per image below which the Winograd output / adjoint input transforms leave the BatchNorm sums to the ordinary
# statistics pass (the sum-reducing kernels need one block per (channel, image); tests set 0 to exercise them on small shapes)
WINO_STATS_MIN_TILES = 256


def wino_down(U, V, pl, stats=False, out_dtype=torch.float32):
    """lo = Out(U * V) stored as `out_dtype`; stats=True: -> (lo, StatParts of lo) — the output transform also reduces the
    BatchNorm sums"""
    import ctypes
    M = _buf(pl.nM, V)
    lo = torch.empty((pl.nb, pl.clo, pl.hlo, pl.wlo), dtype=out_dtype, device=V.device)
    es = lo.element_size()
    if pl.split:
        _call("wfae_wino_gemm_down_split", pl.gemm_flops, 2 * pl.planes * (pl.nU + pl.nV) + 4 * pl.nM, pl.variant, U.data_ptr(),
              V.data_ptr(), _p(M), pl.planes, *pl.dims, _stream(), label="wfae_wino_gemm_down")
    else:
        _call("wfae_wino_gemm_down", pl.gemm_flops, 4 * (pl.nU + pl.nV + pl.nM), pl.variant, _p(U), _p(V), _p(M), *pl.dims, _stream())
    m = 4 if pl.variant else 2
    # images of fewer than 256 tiles: the transform runs one thread per (channel, image, tile) (full waves, whole cache
    # lines of M) and the sums of the small result are taken by the ordinary statistics pass
    small = (pl.hlo // m) * (pl.wlo // m) < WINO_STATS_MIN_TILES
    if not stats or small:
        if out_dtype == BF16:
            _call("wfae_wino_out_bf16", 0, 4 * pl.nM + es * lo.numel(), pl.variant, _p(M), _p(lo), pl.nb, pl.clo, pl.hlo, pl.wlo,
                  None, 0, None, _stream(), label="wfae_wino_out")
        else:
            _call("wfae_wino_out", 0, 4 * (pl.nM + lo.numel()), pl.variant, _p(M), _p(lo), pl.nb, pl.clo, pl.hlo, pl.wlo, _stream())
        return (lo, None) if stats else lo
    cap = 2 * pl.clo * pl.nb * (((pl.hlo // m) * (pl.wlo // m) + 255) // 256)
    part = torch.empty(cap, dtype=torch.float64, device=V.device)
    splits = ctypes.c_int(0)
    _call("wfae_wino_out_bf16" if out_dtype == BF16 else "wfae_wino_out_stats", 0, 4 * pl.nM + es * lo.numel(), pl.variant, _p(M),
          _p(lo), pl.nb, pl.clo, pl.hlo, pl.wlo, part.data_ptr(), cap, ctypes.cast(ctypes.pointer(splits), ctypes.c_void_p),
          _stream(), label="wfae_wino_out")
    return lo, StatParts(part, splits.value)


def wino_up(U, Mt, pl, stats=False, out_dtype=torch.float32):
    """hi = In^T(U^T * Mt) stored as `out_dtype`; stats=True: -> (hi, StatParts of hi)"""
    import ctypes
    dV = _buf(pl.nV, Mt)
    hi = torch.empty((pl.nb, pl.chi, 2 * pl.hlo, 2 * pl.wlo), dtype=out_dtype, device=Mt.device)
    es = hi.element_size()
    if pl.split:
        _call("wfae_wino_gemm_up_split", pl.gemm_flops, 2 * pl.planes * (pl.nU + pl.nM) + 4 * pl.nV, pl.variant,
              U.data_ptr() + 2 * pl.planes * pl.nU, Mt.data_ptr(), _p(dV), pl.planes, *pl.dims, _stream(), label="wfae_wino_gemm_up")
    else:
        _call("wfae_wino_gemm_up", pl.gemm_flops, 4 * (pl.nU + pl.nV + pl.nM), pl.variant, _p(U), _p(Mt), _p(dV), *pl.dims, _stream())
    m = 4 if pl.variant else 2
    small = (pl.hlo // m) * (pl.wlo // m) < WINO_STATS_MIN_TILES
    if not stats or small:
        if out_dtype == BF16:
            _call("wfae_wino_in_t_bf16", 0, 4 * pl.nV + es * hi.numel(), pl.variant, _p(dV), _p(hi), pl.nb, pl.chi, pl.hlo, pl.wlo,
                  None, 0, None, _stream(), label="wfae_wino_in_t")
        else:
            _call("wfae_wino_in_t", 0, 4 * (pl.nV + hi.numel()), pl.variant, _p(dV), _p(hi), pl.nb, pl.chi, pl.hlo, pl.wlo, _stream())
        return (hi, None) if stats else hi
    cap = 2 * pl.chi * pl.nb * (((pl.hlo // m) * (pl.wlo // m) + 255) // 256)
    part = torch.empty(cap, dtype=torch.float64, device=Mt.device)
    splits = ctypes.c_int(0)
    _call("wfae_wino_in_t_bf16" if out_dtype == BF16 else "wfae_wino_in_t_stats", 0, 4 * pl.nV + es * hi.numel(), pl.variant, _p(dV),
          _p(hi), pl.nb, pl.chi, pl.hlo, pl.wlo, part.data_ptr(), cap, ctypes.cast(ctypes.pointer(splits), ctypes.c_void_p),
          _stream(), label="wfae_wino_in_t")
    return hi, StatParts(part, splits.value)


def wino_wgrad(Mt, V, dw, pl, accumulate=False):
    """dw (Clo,Chi,4,4) (+)= G^T (Mt * V^T) G"""
    _chk(dw)
    if pl.split:
        ws = workspace(min(pl.nU * 4 * 13, max(pl.nU * 4 * 6, 1 << 30)))
        _call("wfae_wino_gemm_wgrad_split", pl.gemm_flops, 2 * pl.planes * (pl.nV + pl.nM) + 4 * pl.nU, pl.variant, Mt.data_ptr(),
              V.data_ptr(), _p(dw), pl.planes, *pl.dims, int(accumulate), ws.data_ptr(), ws.numel(), _stream(),
              label="wfae_wino_gemm_wgrad")
        return dw
    ws = workspace(pl.nU * 4 * 6)
    _call("wfae_wino_gemm_wgrad", pl.gemm_flops, 4 * (pl.nU + pl.nV + pl.nM), pl.variant, _p(Mt), _p(V), _p(dw), *pl.dims,
          int(accumulate), ws.data_ptr(), ws.numel(), _stream())
    return dw


def conv4x4s2_down(hi, w):
    """hi (N,Chi,2H,2W), w (Clo,Chi,4,4) -> lo (N,Clo,H,W)"""
    _chka(hi)
    _chk(w)
    nb, chi, h2, w2 = hi.shape
    clo = w.shape[0]
    hlo, wlo = h2 // 2, w2 // 2
    pl = wino_plan(nb, chi, clo, hlo, wlo)
    if pl is not None:
        return wino_down(wino_weights(w, pl), wino_in(hi, pl), pl, out_dtype=hi.dtype)
    if hi.dtype == BF16:      # shapes without a Winograd form run the fp32 gather GEMM
        return to_bf16(conv4x4s2_down(to_f32(hi), w))
    lo = torch.empty((nb, clo, hlo, wlo), dtype=hi.dtype, device=hi.device)
    _call("wfae_conv4x4s2_down", 32 * nb * hlo * wlo * clo * chi, 4 * (nb * hlo * wlo * (clo + 4 * chi) + 16 * clo * chi), _p(hi), _p(w), _p(lo), nb, chi, clo, hlo, wlo, _stream())
    return lo


def conv4x4s2_up(lo, w):
    """lo (N,Clo,H,W), w (Clo,Chi,4,4) -> hi (N,Chi,2H,2W)"""
    _chka(lo)
    _chk(w)
    nb, clo, hlo, wlo = lo.shape
    chi = w.shape[1]
    pl = wino_plan(nb, chi, clo, hlo, wlo)
    if pl is not None:
        return wino_up(wino_weights(w, pl), wino_out_t(lo, pl), pl, out_dtype=lo.dtype)
    if lo.dtype == BF16:
        return to_bf16(conv4x4s2_up(to_f32(lo), w))
    hi = torch.empty((nb, chi, 2 * hlo, 2 * wlo), dtype=lo.dtype, device=lo.device)
    ws = workspace(w.numel() * 4)
    _call("wfae_conv4x4s2_up", 32 * nb * hlo * wlo * clo * chi, 4 * (nb * hlo * wlo * (clo + 4 * chi) + 16 * clo * chi), _p(lo), _p(w), _p(hi), nb, chi, clo, hlo, wlo, ws.data_ptr(), ws.numel(), _stream())
    return hi


def conv4x4s2_wgrad(lo, hi, dw, accumulate=False):
    _chk(dw)
    if wino_plan(lo.shape[0], hi.shape[1], lo.shape[1], lo.shape[2], lo.shape[3]) is None:
        lo, hi = to_f32(lo), to_f32(hi)
        _chk(lo, hi)
    nb, clo, hlo, wlo = lo.shape
    chi = hi.shape[1]
    pl = wino_plan(nb, chi, clo, hlo, wlo)
    if pl is not None:
        return wino_wgrad(wino_out_t(lo, pl), wino_in(hi, pl), dw, pl, accumulate)
    ws = workspace(dw.numel() * 4 * 2)
    _call("wfae_conv4x4s2_wgrad", 32 * nb * hlo * wlo * clo * chi, 4 * (nb * hlo * wlo * (clo + 4 * chi) + 16 * clo * chi), _p(lo), _p(hi), _p(dw), nb, chi, clo, hlo, wlo, int(accumulate),
              ws.data_ptr(), ws.numel(), _stream())
    return dw


# ------------------------------------------------------------- direct convs
def dconv_fwd(x, w, bias, ks, stride, pad, groups, out_dtype=torch.float32):
    """direct convolution; bf16 storage serves the two full-resolution layers with one channel on one side: the first layer
    (x fp32, one channel -> out_dtype bf16) and the output convolution (x bf16 -> fp32)"""
    _chk(w, bias)
    nb, cin, h, wd = x.shape
    cout = w.shape[0]
    ho = (h + 2 * pad - ks) // stride + 1
    wo = (wd + 2 * pad - ks) // stride + 1
    fl, n_in, n_out = 2 * nb * ho * wo * cout * (cin // groups) * ks * ks, x.numel(), nb * cout * ho * wo
    if x.dtype == BF16:
        _chka(x)
        if not (ks == 3 and stride == 1 and pad == 1 and groups == 1 and out_dtype == torch.float32):
            raise _lib.WfaeError("dconv_fwd: a bf16-stored input is served for the 3x3 stride-1 output convolution only")
        y = torch.empty((nb, cout, ho, wo), dtype=torch.float32, device=x.device)
        _call("wfae_dconv_fwd_bf16in", fl, 2 * n_in + 4 * n_out + 4 * w.numel(), _p(x), _p(w), _p(bias), _p(y), nb, cin, cout, h, wd,
              _stream(), label="wfae_dconv_fwd")
        return y
    _chk(x)
    if out_dtype == BF16:
        if not (cin == 1 and groups == 1 and ks == 4 and stride == 2 and pad == 1):
            raise _lib.WfaeError("dconv_fwd: a bf16-stored result is served for the one-channel 4x4 stride-2 first layer only")
        y = torch.empty((nb, cout, ho, wo), dtype=BF16, device=x.device)
        _call("wfae_dconv_fwd_bf16out", fl, 4 * n_in + 2 * n_out + 4 * w.numel(), _p(x), _p(w), _p(bias), _p(y), nb, cout, h, wd,
              _stream(), label="wfae_dconv_fwd")
        return y
    y = torch.empty((nb, cout, ho, wo), dtype=x.dtype, device=x.device)
    _call("wfae_dconv_fwd", 2 * nb * ho * wo * cout * (cin // groups) * ks * ks, 4 * (nb * (cin * h * wd + cout * ho * wo) + w.numel()), _p(x), _p(w), _p(bias), _p(y), nb, cin, cout, h, wd, ks, stride, pad, groups, _stream())
    return y


def dconv_bwd_data(dy, w, cin, ks, pad, groups, out_dtype=torch.float32):
    """data gradient of a stride-1 convolution; the input plane is (Ho + ks - 1 - 2 pad)^2.  out_dtype bf16: the output
    convolution's data gradient (dy fp32, one channel)"""
    _chk(dy, w)
    nb, cout, ho, wo = dy.shape
    h, wd = ho + ks - 1 - 2 * pad, wo + ks - 1 - 2 * pad
    if out_dtype == BF16:
        if not (cout == 1 and groups == 1 and ks == 3 and pad == 1):
            raise _lib.WfaeError("dconv_bwd_data: a bf16-stored result is served for the one-channel output convolution only")
        dx = torch.empty((nb, cin, h, wd), dtype=BF16, device=dy.device)
        _call("wfae_dconv_bwd_data_bf16out", 2 * nb * h * wd * cin * 9, 4 * dy.numel() + 2 * dx.numel(), _p(dy), _p(w), _p(dx), nb,
              cin, h, wd, _stream(), label="wfae_dconv_bwd_data")
        return dx
    dx = torch.empty((nb, cin, h, wd), dtype=dy.dtype, device=dy.device)
    _call("wfae_dconv_bwd_data", 2 * nb * h * wd * cout * (cin // groups) * ks * ks, 4 * (nb * h * wd * (cin + cout) + w.numel()), _p(dy), _p(w), _p(dx), nb, cin, cout, h, wd, ks, pad, groups, _stream())
    return dx


def dconv_bwd_weight(dy, x, dw, ks, stride, pad, groups, accumulate=False):
    _chk(dw)
    nb, cin, h, wd = x.shape
    cout = dy.shape[1]
    ws = workspace()
    if dy.dtype == BF16 or x.dtype == BF16:
        # the two layers with one channel on one side: the big tensor is bf16, the one-channel tensor fp32
        first = dy.dtype == BF16 and x.dtype == torch.float32 and cin == 1 and ks == 4 and stride == 2 and pad == 1
        last = x.dtype == BF16 and dy.dtype == torch.float32 and cout == 1 and ks == 3 and stride == 1 and pad == 1
        if not ((first or last) and groups == 1):
            raise _lib.WfaeError("dconv_bwd_weight: bf16 storage serves the first layer and the output convolution only")
        big, small, c = (dy, x, cout) if first else (x, dy, cin)
        _chka(big)
        _chk(small)
        _call("wfae_c1_wgrad_bf16", 2 * dy.numel() * cin * ks * ks, 2 * big.numel() + 4 * small.numel(), 0 if first else 1, _p(big),
              _p(small), _p(dw), nb, c, h, wd, int(accumulate), ws.data_ptr(), ws.numel(), _stream(), label="wfae_dconv_bwd_weight")
        return dw
    _chk(dy, x)
    _call("wfae_dconv_bwd_weight", 2 * dy.numel() * (cin // groups) * ks * ks, 4 * (x.numel() + dy.numel() + dw.numel()), _p(dy), _p(x), _p(dw), nb, cin, cout, h, wd, ks, stride, pad, groups,
              int(accumulate), ws.data_ptr(), ws.numel(), _stream())
    return dw


def conv4x4s1_supported(c_read):
    """MFMA path of the stride-1 4x4 convolution: the channel count of the operand that is read must be a
    multiple of 16 (smaller layers go through the direct kernels)"""
    return c_read % 16 == 0


def conv4x4s1_fwd(x, w, pad=1, transposed=False):
    """4x4 stride-1 convolution on the GEMM kernel; transposed=True: x is dy, the result is the data gradient"""
    _chk(x, w)
    nb, c, h, wd = x.shape
    cout, cin = w.shape[0], w.shape[1]
    pe = 3 - pad if transposed else pad
    m = cin if transposed else cout
    ho, wo = h + 2 * pe - 3, wd + 2 * pe - 3
    y = torch.empty((nb, m, ho, wo), dtype=x.dtype, device=x.device)
    ws = workspace()
    _call("wfae_conv4x4s1_fwd", 32 * nb * ho * wo * cin * cout, 4 * (x.numel() + y.numel() + w.numel()), _p(x), _p(w), _p(y),
          nb, cin, cout, h, wd, pad, int(transposed), ws.data_ptr(), ws.numel(), _stream())
    return y


def conv4x4s1_bwd_weight(dy, x, dw, pad=1, accumulate=False):
    _chk(dy, x, dw)
    nb, cin, h, wd = x.shape
    cout = dy.shape[1]
    ws = workspace()
    _call("wfae_conv4x4s1_bwd_weight", 32 * dy.numel() * cin, 4 * (x.numel() + dy.numel() + dw.numel()), _p(dy), _p(x), _p(dw),
          nb, cin, cout, h, wd, pad, int(accumulate), ws.data_ptr(), ws.numel(), _stream())
    return dw


def gconv3x3_supported(c, groups):
    return c % groups == 0 and (c // groups) in (4, 8, 16, 32)


_G3B = True   # A/B through set_g3b()


def set_g3b(on):
    """A/B switch: bf16 grouped 3x3 convolutions on the implicit-GEMM kernel (csrc/g3b.hip) or on the dconv.hip kernels"""
    global _G3B
    _G3B = bool(on)


def g3b_supported(c, h, wd, groups):
    return _G3B and _lib.load().wfae_get_matmul_precision() == 1 and bool(
        _lib.load().wfae_g3b_supported(int(c), int(h), int(wd), int(groups)))


def gconv3x3_fwd(x, w, groups, transposed=False):
    """grouped 3x3 'same' conv with Cin == Cout (Bottleneck middle conv); transposed=True gives the data gradient."""
    sfx, es = _chka(x)
    _chk(w)
    nb, c, h, wd = x.shape
    y = torch.empty_like(x)
    ws = workspace()
    cpg = c // groups
    if sfx == "" and _G3B and _lib.load().wfae_g3b_f32_supported(int(c), int(h), int(wd), int(groups), 0):
        _call("wfae_g3b_fwd", 2 * x.numel() * cpg * 9, 2 * es * x.numel(), _p(x), _p(w), _p(y), nb, c, h, wd, groups,
              int(transposed), ws.data_ptr(), ws.numel(), _stream(), label="wfae_gconv3x3_fwd", peak=PEAK_BF16_MFMA / 6)
        return y
    if sfx == "_bf16" and g3b_supported(c, h, wd, groups):
        _call("wfae_g3b_fwd_bf16", 2 * x.numel() * cpg * 9, 2 * es * x.numel(), _p(x), _p(w), _p(y), nb, c, h, wd, groups,
              int(transposed), ws.data_ptr(), ws.numel(), _stream(), label="wfae_gconv3x3_fwd", peak=PEAK_BF16_MFMA)
        return y
    _call("wfae_gconv3x3_fwd" + sfx, 2 * x.numel() * cpg * 9, 2 * es * x.numel(), _p(x), _p(w), _p(y), nb, c, h, wd, groups,
          int(transposed), ws.data_ptr(), ws.numel(), _stream(), label="wfae_gconv3x3_fwd")
    return y


def gconv3x3_bwd_weight(dy, x, dw, groups, accumulate=False):
    sfx, es = _chka(dy, x)
    _chk(dw)
    nb, c, h, wd = x.shape
    ws = workspace()
    if sfx == "" and _G3B and _lib.load().wfae_g3b_f32_supported(int(c), int(h), int(wd), int(groups), 1):
        _call("wfae_g3b_bwd_weight", 2 * x.numel() * (c // groups) * 9, 2 * es * x.numel(), _p(dy), _p(x), _p(dw), nb, c, h, wd,
              groups, int(accumulate), ws.data_ptr(), ws.numel(), _stream(), label="wfae_gconv3x3_bwd_weight", peak=PEAK_BF16_MFMA / 6)
        return dw
    if sfx == "_bf16" and g3b_supported(c, h, wd, groups):
        _call("wfae_g3b_bwd_weight_bf16", 2 * x.numel() * (c // groups) * 9, 2 * es * x.numel(), _p(dy), _p(x), _p(dw), nb, c, h, wd,
              groups, int(accumulate), ws.data_ptr(), ws.numel(), _stream(), label="wfae_gconv3x3_bwd_weight", peak=PEAK_BF16_MFMA)
        return dw
    _call("wfae_gconv3x3_bwd_weight" + sfx, 2 * x.numel() * (c // groups) * 9, 2 * es * x.numel(), _p(dy), _p(x), _p(dw), nb, c,
          h, wd, groups, int(accumulate), ws.data_ptr(), ws.numel(), _stream(), label="wfae_gconv3x3_bwd_weight")
    return dw


# ---------------------------------------------------------------- BatchNorm
class BnStats:
    """per-channel vectors produced by the statistics kernels"""
    __slots__ = ("mean", "invstd", "scale", "shift")

    def __init__(self, c, device):
        buf = torch.empty((4, c), dtype=torch.float32, device=device)
        self.mean, self.invstd, self.scale, self.shift = buf[0], buf[1], buf[2], buf[3]


def bn_stats_train(x, gamma, beta, running_mean, running_var, eps=1e-5, momentum=0.1):
    sfx, es = _chka(x)
    _chk(gamma, beta, running_mean, running_var)
    nb, c, h, wd = x.shape
    st = BnStats(c, x.device)
    ws = workspace()
    _call("wfae_bn_stats_train" + sfx, 0, es * x.numel(), _p(x), nb, c, h * wd, _p(gamma), _p(beta), eps, momentum, _p(running_mean),
              _p(running_var), _p(st.mean), _p(st.invstd), _p(st.scale), _p(st.shift), ws.data_ptr(), ws.numel(),
              _stream(), label="wfae_bn_stats_train")
    return st


def bn_stats_from_rows(sr, shape, gamma, beta, running_mean, running_var, eps=1e-5, momentum=0.1):
    """bn_stats_train of a tensor of `shape` (N, C, H, W) whose partial sums a producer epilogue left in `sr`"""
    _chk(gamma, beta, running_mean, running_var)
    nb, c, h, wd = shape
    st = BnStats(c, gamma.device)
    ws = workspace()
    _call("wfae_bn_stats_from_rows", 0, 8 * sr.part.numel(), sr.part.data_ptr(), sr.rows, nb, c, h * wd, _p(gamma), _p(beta), eps,
          momentum, _p(running_mean), _p(running_var), _p(st.mean), _p(st.invstd), _p(st.scale), _p(st.shift),
          ws.data_ptr(), ws.numel(), _stream(), label="wfae_bn_stats_train")
    return st


def bn_stats_from_parts(sp, shape, gamma, beta, running_mean, running_var, eps=1e-5, momentum=0.1):
    """bn_stats_train of a tensor of `shape` (N, C, H, W) whose fp64 partial sums a producer kernel left in `sp`"""
    _chk(gamma, beta, running_mean, running_var)
    nb, c, h, wd = shape
    st = BnStats(c, gamma.device)
    _call("wfae_bn_stats_from_parts", 0, 8 * sp.part.numel(), sp.part.data_ptr(), sp.splits, nb, c, h * wd, _p(gamma), _p(beta),
          eps, momentum, _p(running_mean), _p(running_var), _p(st.mean), _p(st.invstd), _p(st.scale), _p(st.shift), _stream(),
          label="wfae_bn_stats_train")
    return st


def bn_act_fwd_stats(x, st, act=1):
    """bn_act_fwd that also reduces the BatchNorm sums of its OUTPUT -> (y, StatParts)"""
    import ctypes
    sfx, es = _chka(x)
    nb, c, h, wd = x.shape
    y = torch.empty_like(x)
    # capacity for the library's LARGEST split count per (image, channel) — its scalar path, cdiv(HW, 1024) clamped to
    # 1..1024 (the 16-byte path, taken by pointer alignment as well as by shape, needs fewer); `splits` comes back from the call
    cap = 2 * c * nb * max(1, min(1024, (h * wd + 1023) // 1024))
    part = torch.empty(cap, dtype=torch.float64, device=x.device)
    splits = ctypes.c_int(0)
    _call("wfae_bn_act_fwd_stats" + sfx, 0, 2 * es * x.numel(), _p(x), _p(st.scale), _p(st.shift), _p(y), nb, c, h * wd, act,
          part.data_ptr(), cap, ctypes.cast(ctypes.pointer(splits), ctypes.c_void_p), _stream(), label="wfae_bn_act_fwd")
    return y, StatParts(part, splits.value)


def bn_fold_eval(gamma, beta, running_mean, running_var, eps=1e-5):
    _chk(gamma, beta, running_mean, running_var)
    c = gamma.shape[0]
    st = BnStats(c, gamma.device)
    _call("wfae_bn_fold_eval", 0, 0, _p(gamma), _p(beta), _p(running_mean), _p(running_var), eps, _p(st.mean),
              _p(st.invstd), _p(st.scale), _p(st.shift), c, _stream())
    return st


def bn_act_fwd(x, st, act=1):
    sfx, es = _chka(x)
    nb, c, h, wd = x.shape
    y = torch.empty_like(x)
    _call("wfae_bn_act_fwd" + sfx, 0, 2 * es * x.numel(), _p(x), _p(st.scale), _p(st.shift), _p(y), nb, c, h * wd, act, _stream(),
          label="wfae_bn_act_fwd")
    return y


def bn_act_bwd(dy, x, gamma, st, dgamma, dbeta, res=None, act=1, training=True, accumulate=False, need_dx=True):
    sfx, es = _chka(dy, x, res)
    _chk(gamma, dgamma, dbeta)
    nb, c, h, wd = x.shape
    dx = torch.empty_like(x) if need_dx else None
    ws = workspace()
    args = (_p(dy), _p(x), _p(gamma), _p(st.scale), _p(st.shift), _p(st.mean), _p(st.invstd), _p(res), _p(dx), _p(dgamma),
            _p(dbeta), nb, c, h * wd, act, int(training), int(accumulate))
    tail = (ws.data_ptr(), ws.numel(), _stream())
    n = x.numel()
    if _prof is None or not need_dx:
        _call("wfae_bn_act_bwd" + sfx, 0, es * n * (6 if res is not None else 5), *args, 3 if need_dx else 1, *tail,
              label="wfae_bn_act_bwd")
    else:
        # kernel-granular timing for the roofline read-out: the two kernels of this entry point separately
        _call("wfae_bn_act_bwd" + sfx, 0, 2 * es * n, *args, 1, *tail, label="wfae_bn_act_bwd[reduce]")
        _call("wfae_bn_act_bwd" + sfx, 0, es * n * (4 if res is not None else 3), *args, 2, *tail, label="wfae_bn_act_bwd[dx]")
    return dx


# ------------------------------------------------------------- element-wise
def _ew(name, a, b=None):
    _chk(a, b)
    out = torch.empty_like(a)
    if b is None:
        _call(name, 0, 8 * a.numel(), _p(a), _p(out), a.numel(), _stream())
    else:
        _call(name, 0, 12 * a.numel(), _p(a), _p(b), _p(out), a.numel(), _stream())
    return out


def gelu_fwd(x):
    return _ew("wfae_gelu_fwd", x)


def gelu_bwd(dy, x):
    return _ew("wfae_gelu_bwd", dy, x)


def sigmoid_fwd(x):
    return _ew("wfae_sigmoid_fwd", x)


def sigmoid_bwd(dy, y):
    return _ew("wfae_sigmoid_bwd", dy, y)


def add(a, b):
    return _ew("wfae_add", a, b)


def leaky_relu_fwd(x):
    return _ew("wfae_leaky_relu_fwd", x)


def leaky_relu_bwd(dy, x):
    return _ew("wfae_leaky_relu_bwd", dy, x)


def pad2d(x, pad):
    _chk(x)
    nb, c, h, wd = x.shape
    y = torch.empty((nb, c, h + 2 * pad, wd + 2 * pad), dtype=x.dtype, device=x.device)
    _call("wfae_pad2d", 0, 4 * (x.numel() + y.numel()), _p(x), _p(y), nb * c, h, wd, pad, 0, _stream())
    return y


def crop2d(xp, pad):
    _chk(xp)
    nb, c, hp, wp = xp.shape
    y = torch.empty((nb, c, hp - 2 * pad, wp - 2 * pad), dtype=xp.dtype, device=xp.device)
    _call("wfae_pad2d", 0, 4 * (xp.numel() + y.numel()), _p(xp), _p(y), nb * c, hp - 2 * pad, wp - 2 * pad, pad, 1, _stream())
    return y


def mean_fwd(x, hinge=False, sign=1.0, weight=1.0):
    """weight * mean(x)  or  weight * mean(relu(1 + sign*x))"""
    _chk(x)
    out = torch.empty((), dtype=torch.float32, device=x.device)
    ws = workspace()
    _call("wfae_mean_fwd", 0, 4 * x.numel(), _p(x), _p(out), x.numel(), int(hinge), sign, weight, ws.data_ptr(), ws.numel(), _stream())
    return out


def mean_bwd(x, gout, hinge=False, sign=1.0, weight=1.0):
    _chk(x, gout)
    dx = torch.empty_like(x)
    _call("wfae_mean_bwd", 0, 8 * x.numel(), _p(x), _p(gout), _p(dx), x.numel(), int(hinge), sign, weight, _stream())
    return dx


def scale(x, s=1.0, s_dev=None, out=None):
    """x * s * (s_dev[0] if given); out may alias x"""
    _chk(x, s_dev)
    y = torch.empty_like(x) if out is None else out
    _call("wfae_scale", 0, 8 * x.numel(), _p(x), _p(s_dev), float(s), _p(y), x.numel(), _stream())
    return y


def reduce_sum(x, outer, c, inner, out, accumulate=False):
    _chk(x, out)
    ws = workspace()
    _call("wfae_reduce_sum", 0, 4 * x.numel(), _p(x), outer, c, inner, _p(out), int(accumulate), ws.data_ptr(), ws.numel(), _stream())
    return out


# --------------------------------------------------------------------- loss
def sigmoid_l1_fwd(h, x, weight=1.0):
    _chk(h, x)
    recon = torch.empty_like(h)
    loss = torch.empty((), dtype=torch.float32, device=h.device)
    ws = workspace()
    _call("wfae_sigmoid_l1_fwd", 0, 12 * h.numel(), _p(h), _p(x), _p(recon), _p(loss), weight, h.numel(), ws.data_ptr(), ws.numel(),
              _stream())
    return recon, loss


def sigmoid_l1_bwd(recon, x, gloss, weight=1.0):
    _chk(recon, x, gloss)
    dh = torch.empty_like(recon)
    _call("wfae_sigmoid_l1_bwd", 0, 12 * recon.numel(), _p(recon), _p(x), _p(gloss), weight, _p(dh), recon.numel(), _stream())
    return dh


def l1_fwd(recon, x, weight=1.0):
    _chk(recon, x)
    loss = torch.empty((), dtype=torch.float32, device=recon.device)
    ws = workspace()
    _call("wfae_l1_fwd", 0, 8 * recon.numel(), _p(recon), _p(x), _p(loss), weight, recon.numel(), ws.data_ptr(), ws.numel(), _stream())
    return loss


def l1_bwd(recon, x, gloss, weight=1.0):
    _chk(recon, x, gloss)
    d = torch.empty_like(recon)
    _call("wfae_l1_bwd", 0, 12 * recon.numel(), _p(recon), _p(x), _p(gloss), weight, _p(d), recon.numel(), _stream())
    return d


# ------------------------------------------------- Path-B latent forecaster
def latent_diff_pack(v, tin):
    """v (B,T,C,H,W) -> X (B*H*W, tin*C), Y (B*H*W, (T-tin)*C), both differenced against frame tin-1"""
    _chk(v)
    b, t, c, h, w = v.shape
    X = torch.empty((b * h * w, tin * c), dtype=torch.float32, device=v.device)
    Y = torch.empty((b * h * w, (t - tin) * c), dtype=torch.float32, device=v.device)
    _call("wfae_latent_diff_pack", 0, 8 * v.numel(), _p(v), _p(X), _p(Y), b, t, tin, c, h * w, _stream())
    return X, Y


def latent_unpack_add(pred, v, tin):
    """pred (B*H*W, Tout*C) + last input frame of v (B,T,C,H,W) -> (B,Tout,C,H,W)"""
    _chk(pred, v)
    b, t, c, h, w = v.shape
    out = torch.empty((b, t - tin, c, h, w), dtype=torch.float32, device=v.device)
    _call("wfae_latent_unpack_add", 0, 8 * out.numel(), _p(pred), _p(v), _p(out), b, t, tin, c, h * w, _stream())
    return out


def mse_fwd(pred, target):
    _chk(pred, target)
    loss = torch.empty((), dtype=torch.float32, device=pred.device)
    ws = workspace()
    _call("wfae_mse_fwd", 0, 8 * pred.numel(), _p(pred), _p(target), _p(loss), pred.numel(), ws.data_ptr(), ws.numel(), _stream())
    return loss


def mse_bwd(pred, target, gloss):
    _chk(pred, target, gloss)
    d = torch.empty_like(pred)
    _call("wfae_mse_bwd", 0, 12 * pred.numel(), _p(pred), _p(target), _p(gloss), _p(d), pred.numel(), _stream())
    return d


def ssim_fwd(x, y, clamp01=False):
    _chk(x, y)
    nb = x.shape[0] * x.shape[1]
    h, wd = x.shape[-2:]
    out = torch.empty((), dtype=torch.float32, device=x.device)
    ws = workspace()
    _call("wfae_ssim_fwd", 0, 8 * x.numel(), _p(x), _p(y), _p(out), nb, h, wd, int(clamp01), ws.data_ptr(), ws.numel(), _stream())
    return out


def ssim_bwd(x, y, gout):
    _chk(x, y, gout)
    nb = x.shape[0] * x.shape[1]
    h, wd = x.shape[-2:]
    dy = torch.empty_like(y)
    ws = workspace(3 * nb * (h - 10) * (wd - 10) * 4)
    _call("wfae_ssim_bwd", 0, 12 * x.numel(), _p(x), _p(y), _p(gout), _p(dy), nb, h, wd, ws.data_ptr(), ws.numel(), _stream())
    return dy


def psnr(pred, target, clamp01=False):
    _chk(pred, target)
    nb = pred.shape[0]
    hw = pred.numel() // nb
    out = torch.empty((), dtype=torch.float32, device=pred.device)
    ws = workspace()
    _call("wfae_psnr", 0, 8 * pred.numel(), _p(pred), _p(target), _p(out), nb, hw, int(clamp01), ws.data_ptr(), ws.numel(), _stream())
    return out


# ------------------------------------------------------- latent transformer
def layernorm_fwd(x, res, gamma, beta, eps=1e-5):
    _chk(x, res, gamma, beta)
    rows, e = x.shape
    y = torch.empty_like(x)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    _call("wfae_layernorm_fwd", 0, 12 * x.numel(), _p(x), _p(res), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), rows,
          e, eps, _stream())
    return y, mean, rstd


def layernorm_bwd(dy, x, res, gamma, mean, rstd, dgamma, dbeta, accumulate=False):
    _chk(dy, x, res, gamma, mean, rstd, dgamma, dbeta)
    rows, e = x.shape
    dx = torch.empty_like(x)
    ws = workspace()
    _call("wfae_layernorm_bwd", 0, 16 * x.numel(), _p(dy), _p(x), _p(res), _p(gamma), _p(mean), _p(rstd), _p(dx),
          _p(dgamma), _p(dbeta), rows, e, int(accumulate), ws.data_ptr(), ws.numel(), _stream())
    return dx


def mha_fwd(qkv, s, n, h, d, p_drop=0.0, seed=0, batch_first=False):
    """attention over the s axis for each of n sequences; rows of qkv: s*n+… (seq-first) or n*s+… (batch-first)"""
    _chk(qkv)
    out = torch.empty((s * n, h * d), dtype=torch.float32, device=qkv.device)
    probs = torch.empty((n, h, s, s), dtype=torch.float32, device=qkv.device)
    _call("wfae_mha_fwd", 4 * n * h * s * s * d, 4 * (qkv.numel() + out.numel()), _p(qkv), _p(out), _p(probs),
          s, n, h, d, int(batch_first), p_drop, seed, _stream())
    return out, probs


def mha_bwd(qkv, probs, dout, s, n, h, d, p_drop=0.0, seed=0, batch_first=False):
    _chk(qkv, probs, dout)
    dqkv = torch.empty_like(qkv)
    _call("wfae_mha_bwd", 8 * n * h * s * s * d, 8 * qkv.numel(), _p(qkv), _p(probs), _p(dout), _p(dqkv), s, n,
          h, d, int(batch_first), p_drop, seed, _stream())
    return dqkv


# --------------------------------------------------- AE_ViT_2048 (Path-B)
def patchify(img, patch):
    """(B,C,H,W) -> (B*Hp*Wp, C*P*P)"""
    _chk(img)
    b, c, h, w = img.shape
    hp, wp = h // patch, w // patch
    rows = torch.empty((b * hp * wp, c * patch * patch), dtype=torch.float32, device=img.device)
    _call("wfae_patchify", 0, 8 * img.numel(), _p(img), _p(rows), b, c, hp, wp, patch, _stream())
    return rows


def unpatchify(rows, bias, b, c, hp, wp, patch):
    """(B*Hp*Wp, C*P*P) (+ bias[c]) -> (B,C,Hp*P,Wp*P)"""
    _chk(rows, bias)
    img = torch.empty((b, c, hp * patch, wp * patch), dtype=torch.float32, device=rows.device)
    _call("wfae_unpatchify", 0, 8 * img.numel(), _p(rows), _p(bias), _p(img), b, c, hp, wp, patch, _stream())
    return img


def add_bcast(x, p):
    """x (outer, *inner) + p (*inner)"""
    _chk(x, p)
    out = torch.empty_like(x)
    inner = p.numel()
    _call("wfae_add_bcast", 0, 8 * x.numel(), _p(x), _p(p), _p(out), x.numel() // inner, inner, _stream())
    return out


def copy_rows(src, rows, cols, src_ld, src_off=0, row_div=1, dst_ld=None, dst_off=0, zero_fill=False):
    """dst (rows, dst_ld): dst[r][dst_off + c] = src[(r // row_div) * src_ld + src_off + c] (see wfae_copy_rows)"""
    _chk(src)
    dst_ld = cols if dst_ld is None else dst_ld
    dst = torch.empty((rows, dst_ld), dtype=torch.float32, device=src.device)
    _call("wfae_copy_rows", 0, 4 * rows * (cols + dst_ld), _p(src), _p(dst), rows, cols, src_ld, src_off, row_div, dst_ld,
          dst_off, int(zero_fill), _stream())
    return dst


def sum_mid(x, a, m, bn):
    """x viewed as (a, m, bn) -> (a, bn)"""
    _chk(x)
    out = torch.empty((a, bn), dtype=torch.float32, device=x.device)
    _call("wfae_sum_mid", 0, 4 * x.numel(), _p(x), _p(out), a, m, bn, _stream())
    return out


def sq_attn_fwd(q, kv, b, l, h, d):
    _chk(q, kv)
    out = torch.empty((b, h * d), dtype=torch.float32, device=q.device)
    probs = torch.empty((b, h, l), dtype=torch.float32, device=q.device)
    _call("wfae_sq_attn_fwd", 4 * b * h * l * d, 4 * (kv.numel() + q.numel()), _p(q), _p(kv), _p(out), _p(probs), b, l, h, d, _stream())
    return out, probs


def sq_attn_bwd(q, kv, probs, dout, b, l, h, d):
    _chk(q, kv, probs, dout)
    dq, dkv = torch.empty_like(q), torch.empty_like(kv)
    _call("wfae_sq_attn_bwd", 8 * b * h * l * d, 8 * kv.numel(), _p(q), _p(kv), _p(probs), _p(dout), _p(dq), _p(dkv), b, l, h, d, _stream())
    return dq, dkv


def relu_fwd(x):
    _chk(x)
    y = torch.empty_like(x)
    _call("wfae_relu_fwd", 0, 8 * x.numel(), _p(x), _p(y), x.numel(), _stream())
    return y


def relu_bwd(dy, y):
    _chk(dy, y)
    dx = torch.empty_like(dy)
    _call("wfae_relu_bwd", 0, 12 * y.numel(), _p(dy), _p(y), _p(dx), y.numel(), _stream())
    return dx


def dropout(x, p_drop, seed):
    _chk(x)
    y = torch.empty_like(x)
    _call("wfae_dropout", 0, 8 * x.numel(), _p(x), _p(y), x.numel(), p_drop, seed, _stream())
    return y


# ---------------------------------------------------------------- optimiser
def adamw_(p, g, m, v, lr, beta1, beta2, eps, wd, bc1, bc2, grad_scale=1.0):
    _chk(p, g, m, v)
    _call("wfae_adamw", 0, 28 * p.numel(), _p(p), _p(g), _p(m), _p(v), p.numel(), lr, beta1, beta2, eps, wd, bc1, bc2, grad_scale,
              _stream())


def sumsq(x):
    _chk(x)
    out = torch.empty((), dtype=torch.float64, device=x.device)
    ws = workspace()
    _call("wfae_sumsq", 0, 4 * x.numel(), _p(x), x.numel(), _p(out), ws.data_ptr(), ws.numel(), _stream())
    return out


def sumsq_into(x, out_slot):
    """sum of squares of x written into a 0-dim fp64 view (one slot of a parts vector)"""
    _chk(x)
    ws = workspace()
    _call("wfae_sumsq", 0, 4 * x.numel(), _p(x), x.numel(), _p(out_slot), ws.data_ptr(), ws.numel(), _stream())


def clip_coef(parts, max_norm, pre_scale=1.0):
    """(coef, total_norm) device pair from fp64 partial sums of squares"""
    out = torch.empty(2, dtype=torch.float32, device=parts.device)
    _call("wfae_clip_coef", 0, 0, _p(parts), parts.numel(), float(max_norm), float(pre_scale), _p(out), _stream())
    return out


def adaptive_weight(ss_rec, ss_disc, disc_weight=1.0):
    out = torch.empty((), dtype=torch.float32, device=ss_rec.device)
    _call("wfae_adaptive_weight", 0, 0, _p(ss_rec), _p(ss_disc), float(disc_weight), _p(out), _stream())
    return out


def vil_u8_to_f32(src_nhwt, scale=1.0 / 255.0):
    """uint8 (N,H,W,T) -> fp32 (N,T,H,W) * scale: the loader contract on device."""
    if src_nhwt.dtype != torch.uint8 or not src_nhwt.is_cuda or not src_nhwt.is_contiguous():
        raise _lib.WfaeError("vil_u8_to_f32 needs a contiguous uint8 device tensor")
    n, h, w, t = src_nhwt.shape
    dst = torch.empty((n, t, h, w), dtype=torch.float32, device=src_nhwt.device)
    _call("wfae_vil_u8_to_f32", 0, 5 * dst.numel(), src_nhwt.data_ptr(), _p(dst), n, h, w, t, scale, _stream())
    return dst
