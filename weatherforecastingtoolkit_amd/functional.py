"""autograd glue: torch.autograd.Function wrappers whose forward AND backward are
sequences of libwfae.so kernel launches (ops.py).  Granularity follows the
reference's building blocks so that the fan-out of the residual stream stays
inside one Function (no ATen gradient-accumulation kernels on the hot path):

  DownUnitFn     Conv2d(4,2,1) -> BN -> GELU            EncBlock.down  (ae_64x8x8_lin.py:30-33)
  UpUnitFn       ConvTranspose2d(4,2,1) -> BN -> GELU   DecBlock.up    (:41-44)
  BottleneckFn   x + f(x)                               Bottleneck     (:7-22)
  Conv1x1Fn      1x1 conv (+bias) (+broadcast pos_emb)  enc[4], dec[0] (:69,79,91)
  LinearFn       to_latent / from_latent                (:74-75)
  DConvFn        3x3 'same' conv, any groups            dec[5]         (:84)
  SigmoidFn, L1LossFn, SsimFn                           (:86; experiments/ae_v2/train.py:55,62)

Parameter gradients are written by the kernels straight into the tensor that
`grad_buffer(p)` returns — a view of a flat gradient arena when an optimiser /
data-parallel wrapper registered one (so AdamW and the RCCL all-reduce see one
contiguous buffer), otherwise a fresh tensor — and handed back to autograd,
which adopts it as `p.grad` without a copy.
"""
from __future__ import annotations

import os

import torch
from torch.autograd import Function

from . import ops

BN_EPS = 1e-5


# ---------------------------------------------------------------------------------------------
# Weight-gradient kernels do not feed the backward chain (dgrad -> BN bwd -> dgrad ...), so they
# can run on a second HIP stream: the MFMA-bound wgrad GEMMs then overlap the HBM-bound
# BatchNorm / element-wise backward kernels of the critical path.  Opt-in (bench / train script):
# whoever reads .grad afterwards must call join_side_stream() first (FusedAdamW.step,
# DataParallelTrainer.reduce_gradients and helpers.grad_norm do).
# ---------------------------------------------------------------------------------------------
_overlap = False
_side = {}


# BatchNorm statistics of 1x1-convolution outputs reduced in the GEMM epilogue instead of by a pass over the tensor.
# Round 1 built this with fp32 partial sums (four DPP adds per 64-column wave row) and left it off: the fp32 partials
# are as accurate as the separate pass but not ROUNDED like it, and a 1-ulp difference of a channel mean was enough to
# push a near-tie pixel of the L1 loss across its kink.  The partials are now reduced in fp64 from
# fp32 sums of four — the arithmetic of the separate pass (chan_reduce_kernel) — so the statistics are reproduced to
# fp64 rounding (2e-7 asserted in test_conv1x1_fwd_stats_epilogue).  WFAE_STAT_FUSION=0: separate passes.
STAT_FUSION = os.environ.get("WFAE_STAT_FUSION", "1") == "1"


# BatchNorm-apply + GELU of a Bottleneck's FIRST BatchNorm (the C-channel residual stream: 70 % of the BN/GELU bytes)
# inside the operand loaders of the two GEMMs that consume it (forward C -> C/4 convolution, its weight gradient):
# a1 = gelu(bn1(x)) is never written, saved or re-read (ops.conv1x1_fwd_bnact).  Bit-identical results.
# Measured (B = 32, 384x384, fp32, profiles/r02_v2_*): forward GEMMs + apply passes 21.6 -> 14.8 ms per step, weight
# gradients 12.7 -> 15.0 ms (they evaluate the GELU a second time), step 256.0 -> 249.7 ms, peak memory 106.9 -> 86.4 GiB.
# WFAE_FUSE_A1=0 restores the materialised form, WFAE_FUSE_A1_MAXC limits the fused form to that many input channels (A/B:
# 256 -> 250.1 ms, 512 -> 249.6 ms, all -> 249.7 ms).
FUSE_A1 = os.environ.get("WFAE_FUSE_A1", "1") == "1"
FUSE_A1_MAXC = 1 << 30
# The same for the THIRD BatchNorm of a Bottleneck (C/4 channels -> the C/4 -> C convolution and its weight gradient; the
# roles of that weight gradient are swapped when C/4 < 128, so the prologue sits on its A operand there).  The forward
# GEMM re-loads (and re-activates) every element once per M tile, 1 - 8 times: measured for the whole step, fusing it
# at every width costs 3 ms (247.4 -> 250.5 ms), at C <= 256 0.5 ms, at C <= 128 nothing — so it is limited to
# WFAE_FUSE_A3_MAXC = 128 output channels (one M tile; -1.1 GiB; 256 where csrc/c1r.hip serves the product, below).  WFAE_FUSE_A3=0: off.
FUSE_A3 = os.environ.get("WFAE_FUSE_A3", "1") == "1"
FUSE_A3_MAXC = int(os.environ.get("WFAE_FUSE_A3_MAXC", "128"))
# fp32 tensors on csrc/c1r.hip (round 4): its prologue form of the C/4 -> C product at C = 256 costs 0.519 against 0.507 ms + the
# 0.123 ms apply pass, the weight gradient's prologue sits on the 64-channel operand: the step is unchanged (185.1 / 185.3 against
# 184.9 / 185.4 ms, same box) and a3 is not kept: - 2.25 GiB.  Above that the M-sliced kernels activate the operand once per slice.
# bf16 storage on csrc/c1rb.hip: 93.25 / 92.9 -> 92.68 / 92.71 ms, - 1.1 GiB.
FUSE_A3_MAXC_C1R = max(FUSE_A3_MAXC, 256)


# BatchNorm sums produced by the kernel that WRITES the tensor when that kernel is a streaming one (the Winograd output
# transform in front of a unit's BatchNorm; the unit's BatchNorm+GELU pass in front of the first Bottleneck): fp64
# accumulation like the separate statistics pass, so its results are reproduced to fp64 rounding.  WFAE_PRODUCER_STATS=0: off.
PRODUCER_STATS = os.environ.get("WFAE_PRODUCER_STATS", "1") == "1"


def set_wgrad_overlap(flag: bool):
    global _overlap
    _overlap = bool(flag)


def _side_stream():
    dev = torch.cuda.current_device()
    s = _side.get(dev)
    if s is None:
        s = _side[dev] = torch.cuda.Stream(device=dev)
    return s


# Tensors read by side-stream kernels are kept alive (Python references) until the main stream has
# waited for the side stream; only then may the caching allocator hand their memory to later
# main-stream kernels.  (tensor.record_stream() would also be correct but makes the allocator poll
# per-block events and, with the host running several steps ahead, fall back to fresh hipMallocs.)
_keep = []
_keep_bytes = [0]
_KEEP_LIMIT = 24 << 30


def join_side_stream():
    if _side:
        dev = torch.cuda.current_device()
        if dev in _side:
            torch.cuda.current_stream().wait_stream(_side[dev])
    _keep.clear()
    _keep_bytes[0] = 0


def _wgrad(fn, *tensors):
    """run fn() (a weight-gradient launch reading `tensors`) on the side stream when overlap is on"""
    if not _overlap:
        return fn()
    if _force_main[0]:
        # the gradient buffer handed out last will be summed with an earlier contribution by autograd on the
        # main stream: wait for the side stream (the earlier contribution may still be in flight there) and
        # stay on the main stream
        _force_main[0] = False
        join_side_stream()
        return fn()
    side = _side_stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    _keep.extend(tensors)
    _keep_bytes[0] += sum(t.numel() * t.element_size() for t in tensors)
    if _keep_bytes[0] > _KEEP_LIMIT:
        join_side_stream()


_arena_grads = [True]


class detached_grads:
    """Context in which parameter gradients are written to fresh buffers instead of the optimiser's
    gradient arena — for torch.autograd.grad(...) probes whose results must survive later backward
    passes (the adaptive GAN weight, experiments/ae_v2_2/train.py:46-52)."""

    def __enter__(self):
        self.prev = _arena_grads[0]
        _arena_grads[0] = False

    def __exit__(self, *exc):
        _arena_grads[0] = self.prev


_force_main = [False]


def grad_buffer(p):
    """Where a backward kernel writes the gradient of parameter p: p's slot in the optimiser's gradient
    arena when this is the only contribution autograd will see for p in this pass.  When the gradient is
    going to be ADDED to another one — p already has a .grad (gradient accumulation), or p is used more
    than once in the graph (the discriminator on real and fake batches, ae_v2_2/train.py:88-89; the engine
    sums both before AccumulateGrad runs) — it gets a fresh buffer, and the launch that fills it is kept
    on the main stream behind the side stream (`_wgrad`), because autograd performs that addition on the
    main stream."""
    task = torch._C._current_graph_task_id()
    again = task >= 0 and getattr(p, "_wfae_lent_task", None) == task
    p._wfae_lent_task = task
    if p.grad is not None or again:
        _force_main[0] = True
        return torch.empty_like(p)
    v = getattr(p, "_wfae_grad_view", None)
    if v is not None and _arena_grads[0]:
        # a fresh alias: autograd adopts the tensor as p.grad without a copy only
        # when nobody else references the same TensorImpl
        return v.view(v.shape)
    return torch.empty_like(p)


def _bn_stats(x, bn, training):
    """training: batch statistics + running-stat update; eval: fold running stats."""
    if training:
        st = ops.bn_stats_train(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, bn.momentum)
        bn._nbt_pending += 1
        return st
    return ops.bn_fold_eval(bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps)


def _bn_stats_rows(sr, x, bn, training):
    """_bn_stats of x when a producer kernel may already have reduced its per-channel sums (ops.StatRows: fp32 rows of
    a GEMM epilogue; ops.StatParts: fp64 partials of a streaming producer)"""
    if training and isinstance(sr, ops.StatParts):
        st = ops.bn_stats_from_parts(sr, tuple(x.shape), bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps,
                                     bn.momentum)
        bn._nbt_pending += 1
        return st
    if training and sr is not None:
        st = ops.bn_stats_from_rows(sr, tuple(x.shape), bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps,
                                    bn.momentum)
        bn._nbt_pending += 1
        return st
    return _bn_stats(x, bn, training)


def _use_batch_stats(bn):
    return bn.training or bn.running_mean is None


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


# ----------------------------------------------------------- 4x4 s2 units --
def _down_fwd(x, w, unit=False):
    if w.shape[1] < 16:
        # the one-channel first layer of the MODEL's stack (unit=True: DownUnitFn) reads the fp32 frame and writes the stack's
        # activation storage type; the generic Conv2d(.., 4, 2, 1) layer (Conv4x4DownFn) keeps the dtype of its input
        out = ops.activation_dtype() if (unit and x.dtype == torch.float32 and w.shape[1] == 1) else torch.float32
        return ops.dconv_fwd(x, w, None, 4, 2, 1, 1, out_dtype=out)
    return ops.conv4x4s2_down(x, w)


def _down_wgrad(dlo, hi, dw):
    if hi.shape[1] < 16:
        return ops.dconv_bwd_weight(dlo, hi, dw, 4, 2, 1, 1)
    return ops.conv4x4s2_wgrad(dlo, hi, dw)


# Winograd form of the 4x4 stride-2 GEMMs (ops.wino_*): every tensor is transformed ONCE per step.
# The forward keeps the transformed input (V = In(hi) for a Conv2d, Mt = Out^T(lo) for a ConvTranspose2d) and the
# transformed weights U; the backward transforms the incoming gradient once and feeds it to both the data-
# gradient GEMM and the weight-gradient GEMM.
def _down_plan(x, w):
    """plan for Conv2d(Chi -> Clo, 4, s2, p1) on x (N,Chi,2Hlo,2Wlo); None: direct kernels"""
    if w.shape[1] < 16 or (x.shape[2] & 1) or (x.shape[3] & 1):
        return None
    return ops.wino_plan(x.shape[0], w.shape[1], w.shape[0], x.shape[2] // 2, x.shape[3] // 2)


def _down_forward(x, w, stats=False):
    """-> (t, pl, U, V[, StatParts of t]): 4x4 s2 convolution of x; U, V are None on the direct path"""
    pl = _down_plan(x, w)
    if pl is None:
        return (_down_fwd(x, w, unit=True), None, None, None) + ((None,) if stats else ())
    U, V = ops.wino_weights(w, pl), ops.wino_in(x, pl)
    if stats:
        t, sp = ops.wino_down(U, V, pl, stats=True, out_dtype=x.dtype)
        return t, pl, U, V, sp
    return ops.wino_down(U, V, pl, out_dtype=x.dtype), pl, U, V


def _down_backward(dt, x, w, pl, U, V, need_dx, need_dw):
    """gradients of the 4x4 s2 convolution: (dx or None, dw or None); the weight gradient goes to the side stream"""
    dw = dx = None
    if pl is None:
        if need_dw:
            dw = grad_buffer(w)
            _wgrad(lambda: _down_wgrad(dt, x, dw), dt, x)
        if need_dx:
            dx = ops.conv4x4s2_up(dt, w)
        return dx, dw
    Mt = ops.wino_out_t(dt, pl)
    if need_dw:
        dw = grad_buffer(w)
        _wgrad(lambda: ops.wino_wgrad(Mt, V, dw, pl), Mt, V)
    if need_dx:
        dx = ops.wino_up(U, Mt, pl, out_dtype=x.dtype)
    return dx, dw


def _opt(t):
    """placeholder for an absent cached tensor in save_for_backward"""
    return t if t is not None else torch.empty(0)


class DownUnitFn(Function):
    @staticmethod
    def forward(ctx, x, w, gamma, beta, bn):
        x = _c(x)
        training = _use_batch_stats(bn)
        prod = training and PRODUCER_STATS
        t, pl, U, V, sp = _down_forward(x, w, stats=True) if prod else _down_forward(x, w) + (None,)
        st = _bn_stats_rows(sp, t, bn, training)
        if prod:    # the sums of `a` ride along for the BatchNorm of the first Bottleneck (EncBlock.forward hands them on)
            a, bn._out_stats = ops.bn_act_fwd_stats(t, st, 1)
        else:
            a, bn._out_stats = ops.bn_act_fwd(t, st, 1), None
        ctx.save_for_backward(x, w, gamma, t, st.mean, st.invstd, st.scale, st.shift, _opt(U), _opt(V))
        ctx.training, ctx.beta, ctx.pl = training, beta, pl
        return a

    @staticmethod
    def backward(ctx, da):
        x, w, gamma, t, mean, invstd, scale, shift, U, V = ctx.saved_tensors
        st = _mk_stats(mean, invstd, scale, shift)
        dg, db = grad_buffer(gamma), grad_buffer(ctx.beta)
        dt = ops.bn_act_bwd(_c(da), t, gamma, st, dg, db, None, 1, ctx.training)
        dx, dw = _down_backward(dt, x, w, ctx.pl, U, V, ctx.needs_input_grad[0], True)
        return dx, dw, dg, db, None


class UpUnitFn(Function):
    @staticmethod
    def forward(ctx, x, w, gamma, beta, bn):
        x = _c(x)
        training = _use_batch_stats(bn)
        pl = ops.wino_plan(x.shape[0], w.shape[1], w.shape[0], x.shape[2], x.shape[3])
        U = Mt = None
        sp = None
        act_dt = ops.activation_dtype() if x.dtype == torch.float32 else x.dtype   # dec[1] reads the fp32 latent projection
        if pl is not None:
            U, Mt = ops.wino_weights(w, pl), ops.wino_out_t(x, pl)
            if training and PRODUCER_STATS:
                t, sp = ops.wino_up(U, Mt, pl, stats=True, out_dtype=act_dt)
            else:
                t = ops.wino_up(U, Mt, pl, out_dtype=act_dt)
        else:
            t = ops.to_dtype(ops.conv4x4s2_up(x, w), act_dt)
        st = _bn_stats_rows(sp, t, bn, training)
        if training and PRODUCER_STATS:
            a, bn._out_stats = ops.bn_act_fwd_stats(t, st, 1)
        else:
            a, bn._out_stats = ops.bn_act_fwd(t, st, 1), None
        ctx.save_for_backward(x, w, gamma, t, st.mean, st.invstd, st.scale, st.shift, _opt(U), _opt(Mt))
        ctx.training, ctx.beta, ctx.pl = training, beta, pl
        return a

    @staticmethod
    def backward(ctx, da):
        x, w, gamma, t, mean, invstd, scale, shift, U, Mt = ctx.saved_tensors
        st = _mk_stats(mean, invstd, scale, shift)
        dg, db = grad_buffer(gamma), grad_buffer(ctx.beta)
        dt = ops.bn_act_bwd(_c(da), t, gamma, st, dg, db, None, 1, ctx.training)
        dw = grad_buffer(w)
        pl = ctx.pl
        if pl is not None:
            V = ops.wino_in(dt, pl)                                  # the hi-side tensor of this layer is dt
            _wgrad(lambda: ops.wino_wgrad(Mt, V, dw, pl), Mt, V)     # lo = x (Mt kept from forward), hi = dt
            dx = ops.wino_down(U, V, pl, out_dtype=x.dtype) if ctx.needs_input_grad[0] else None
        else:
            _wgrad(lambda: ops.conv4x4s2_wgrad(x, dt, dw), x, dt)    # lo = x, hi = dt
            dx = ops.to_dtype(ops.conv4x4s2_down(dt, w), x.dtype) if ctx.needs_input_grad[0] else None
        return dx, dw, dg, db, None


# ------------------------------------------------- PatchGAN discriminator --
# (pipeline/models/autoencoderkl/losses/model.py:100-150).  Parameters that do not require grad (the
# discriminator during the generator step, Lightning's toggle_optimizer in ae_v2_2/train.py:133) get no
# weight-gradient launches at all.
def _maybe_buffer(p, needed):
    return grad_buffer(p) if needed else torch.empty_like(p)


def _conv4_fwd(x, w, bias, stride):
    if stride == 1:
        if bias is None and ops.conv4x4s1_supported(w.shape[1]):
            return ops.conv4x4s1_fwd(x, w, 1, False)
        return ops.dconv_fwd(x, w, bias, 4, 1, 1, 1)
    if bias is not None or w.shape[1] < 16:
        return ops.dconv_fwd(x, w, bias, 4, 2, 1, 1)
    return ops.conv4x4s2_down(x, w)


def _conv4_wgrad(dy, x, dw, stride):
    if stride == 1:
        if ops.conv4x4s1_supported(x.shape[1]):
            return ops.conv4x4s1_bwd_weight(dy, x, dw, 1)
        return ops.dconv_bwd_weight(dy, x, dw, 4, 1, 1, 1)
    return _down_wgrad(dy, x, dw)


def _conv4_dgrad(dy, w, stride):
    if stride == 1:
        if ops.conv4x4s1_supported(w.shape[0]):
            return ops.conv4x4s1_fwd(dy, w, 1, True)
        return ops.dconv_bwd_data(dy, w, w.shape[1], 4, 1, 1)
    return ops.conv4x4s2_up(dy, w)


class DiscUnitFn(Function):
    """Conv2d(4x4, stride s, pad 1, bias=False) -> BatchNorm2d -> LeakyReLU(0.2)  (model.py:129-141)"""

    @staticmethod
    def forward(ctx, x, w, gamma, beta, bn, stride):
        x = _c(x)
        training = _use_batch_stats(bn)
        pl = U = V = None
        if stride == 2:
            t, pl, U, V = _down_forward(x, w)
        else:
            t = _conv4_fwd(x, w, None, stride)
        st = _bn_stats(t, bn, training)
        a = ops.bn_act_fwd(t, st, 2)
        ctx.save_for_backward(x, w, gamma, t, st.mean, st.invstd, st.scale, st.shift, _opt(U), _opt(V))
        ctx.training, ctx.beta, ctx.stride, ctx.pl = training, beta, stride, pl
        return a

    @staticmethod
    def backward(ctx, da):
        x, w, gamma, t, mean, invstd, scale, shift, U, V = ctx.saved_tensors
        st = _mk_stats(mean, invstd, scale, shift)
        need_w, need_g = ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        dg, db = _maybe_buffer(gamma, need_g), _maybe_buffer(ctx.beta, need_g)
        dt = ops.bn_act_bwd(_c(da), t, gamma, st, dg, db, None, 2, ctx.training)
        if ctx.stride == 2:
            dx, dw = _down_backward(dt, x, w, ctx.pl, U, V, ctx.needs_input_grad[0], need_w)
        else:
            dw = None
            if need_w:
                dw = grad_buffer(w)
                _wgrad(lambda: _conv4_wgrad(dt, x, dw, 1), dt, x)
            dx = _conv4_dgrad(dt, w, 1) if ctx.needs_input_grad[0] else None
        return dx, dw, dg if need_g else None, db if need_g else None, None, None


class Conv4LeakyFn(Function):
    """Conv2d(4x4, stride 2, pad 1, bias) -> LeakyReLU(0.2)  (model.py:125)"""

    @staticmethod
    def forward(ctx, x, w, bias):
        x = _c(x)
        t = _conv4_fwd(x, w, bias, 2)
        ctx.save_for_backward(x, w, t)
        ctx.bias = bias
        return ops.leaky_relu_fwd(t)

    @staticmethod
    def backward(ctx, dy):
        x, w, t = ctx.saved_tensors
        dt = ops.leaky_relu_bwd(_c(dy), t)
        dw = dbias = None
        if ctx.needs_input_grad[1]:
            dw = grad_buffer(w)
            _conv4_wgrad(dt, x, dw, 2)
        if ctx.bias is not None and ctx.needs_input_grad[2]:
            nb, cout, h, wd = dt.shape
            dbias = grad_buffer(ctx.bias)
            ops.reduce_sum(dt, nb, cout, h * wd, dbias)
        dx = ops.conv4x4s2_up(dt, w) if ctx.needs_input_grad[0] else None
        return dx, dw, dbias


class Conv4Fn(Function):
    """Conv2d(4x4, stride 1|2, pad 1, optional bias) as a standalone layer"""

    @staticmethod
    def forward(ctx, x, w, bias, stride):
        x = _c(x)
        ctx.save_for_backward(x, w)
        ctx.bias, ctx.stride = bias, stride
        return _conv4_fwd(x, w, bias, stride)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = _c(dy)
        dw = dbias = None
        if ctx.needs_input_grad[1]:
            dw = grad_buffer(w)
            _conv4_wgrad(dy, x, dw, ctx.stride)
        if ctx.bias is not None and ctx.needs_input_grad[2]:
            nb, cout, h, wd = dy.shape
            dbias = grad_buffer(ctx.bias)
            ops.reduce_sum(dy, nb, cout, h * wd, dbias)
        dx = _conv4_dgrad(dy, w, ctx.stride) if ctx.needs_input_grad[0] else None
        return dx, dw, dbias, None


class Pad2dFn(Function):
    """zero padding of the spatial dims (the `padding=1` of the reference's final 1x1 conv, model.py:145)"""

    @staticmethod
    def forward(ctx, x, pad):
        ctx.pad = pad
        return ops.pad2d(_c(x), pad)

    @staticmethod
    def backward(ctx, dy):
        return ops.crop2d(_c(dy), ctx.pad), None


class LeakyReluFn(Function):
    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        ctx.save_for_backward(x)
        return ops.leaky_relu_fwd(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return ops.leaky_relu_bwd(_c(dy), x)


class MeanFn(Function):
    """weight * mean(x)  or  weight * mean(relu(1 + sign * x))  (hinge terms, contperceptual.py:19-23)"""

    @staticmethod
    def forward(ctx, x, hinge, sign, weight):
        x = _c(x)
        ctx.save_for_backward(x)
        ctx.cfg = (hinge, sign, weight)
        return ops.mean_fwd(x, hinge, sign, weight)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        hinge, sign, weight = ctx.cfg
        return ops.mean_bwd(x, _c(g), hinge, sign, weight), None, None, None


class ScaleByFn(Function):
    """y = x * s with s a detached device scalar (the adaptive weight, ae_v2/train.py:46-52,84)"""

    @staticmethod
    def forward(ctx, x, s):
        ctx.save_for_backward(s)
        return ops.scale(_c(x), 1.0, s)

    @staticmethod
    def backward(ctx, g):
        (s,) = ctx.saved_tensors
        return ops.scale(_c(g), 1.0, s), None


def hinge_d_loss(logits_real, logits_fake):
    """0.5 * (mean(relu(1 - real)) + mean(relu(1 + fake)))  (contperceptual.py:19-23)"""
    return AddFn.apply(MeanFn.apply(logits_real, True, -1.0, 0.5), MeanFn.apply(logits_fake, True, 1.0, 0.5))


def neg_mean(x):
    """generator term -mean(D(x_hat))  (ae_v2/train.py:78)"""
    return MeanFn.apply(x, False, 1.0, -1.0)


# grouped 3x3 conv of the Bottleneck: register-blocked kernels when channels/group is 4/8/16/32
def _g3_fwd(x, w, groups):
    if ops.gconv3x3_supported(x.shape[1], groups) and w.shape[0] == x.shape[1]:
        return ops.gconv3x3_fwd(x, w, groups, False)
    return ops.dconv_fwd(x, w, None, 3, 1, 1, groups)


def _g3_dgrad(dy, w, cin, groups):
    if ops.gconv3x3_supported(cin, groups) and w.shape[0] == cin:
        return ops.gconv3x3_fwd(dy, w, groups, True)
    return ops.dconv_bwd_data(dy, w, cin, 3, 1, groups)


def _g3_wgrad(dy, x, dw, groups):
    c = x.shape[1]
    # measured (tools/kbench.py): pixel-parallel VALU kernel wins at 4 and 8 channels/group, the per-group
    # MFMA implicit GEMM at 16 and 32 (the library picks between the two)
    if c % groups == 0 and c // groups in (4, 8, 16, 32) and dw.shape[0] == c and groups > 1:
        return ops.gconv3x3_bwd_weight(dy, x, dw, groups)
    return ops.dconv_bwd_weight(dy, x, dw, 3, 1, 1, groups)


# -------------------------------------------------------------- Bottleneck --
def _mk_stats(mean, invstd, scale, shift):
    st = ops.BnStats.__new__(ops.BnStats)
    st.mean, st.invstd, st.scale, st.shift = mean, invstd, scale, shift
    return st


def _b16(w, plane, transposed, x, st=None, res=None, stats=False, label="wfae_c1b_fwd"):
    """a Bottleneck 1x1 product on bf16-stored tensors: csrc/c1rb.hip (register-direct, reads the fp32 weight) where it
    serves the shape, else csrc/c1b.hip with the prepared bf16 plane"""
    m = w.shape[1] if transposed else w.shape[0]
    if ops.c1rb_take(m, x.shape[1], x.shape[2] * x.shape[3], st is not None):
        return ops.c1rb_fwd(w, transposed, x, st, res, stats, label)
    return ops.c1b_fwd(plane, x, st, res, stats, label)


def _dgrad_bn(dt, w, Wt, x, gamma, st, dgamma, dbeta, res, training):
    """dx of  conv1x1(gelu(bn(x)), w)  given dt = dL/d(conv output): the data gradient dA = W^T dT followed by the
    BatchNorm + GELU backward at x (+ res, the residual-branch gradient).  bf16-stored tensors: csrc/c1rb.hip, or csrc/c1b.hip
    with Wt, the bf16 plane of w^T; fp32 storage (Wt empty): wfae_conv1x1_bwd_data."""
    if dt.dtype == torch.float32 and ops.c1r_bnred_supported(w.shape[1], dt.shape[1], dt.shape[2] * dt.shape[3]):
        # the C <= 256 widening data gradients on csrc/c1r.hip: sum dU / sum dU xhat of the BatchNorm in front ride in the epilogue
        # (x travels through the kernel's residual ring), the reduce pass over (dA, x) disappears
        # — and so does dA itself: the second pass runs in the epilogue of the same product computed again (ops.c1r_bndx)
        if ops.c1r_bndx_on():
            _, sr = ops.c1r_bnred(w, dt, x, st, store=False)
            ops.bn_act_bwd_from_rows(sr, x.shape[1], dgamma, dbeta)
            return ops.c1r_bndx(w, dt, x, gamma, st, res, training)
        da, sr = ops.c1r_bnred(w, dt, x, st)
        ops.bn_act_bwd_from_rows(sr, x.shape[1], dgamma, dbeta)
        return ops.bn_act_bwd_dx(da, x, gamma, st, res, 1, training)
    if dt.dtype == ops.BF16 and ops.c1rb_supported(w.shape[1], dt.shape[1], dt.shape[2] * dt.shape[3]):
        da = ops.c1rb_fwd(w, True, dt, label="wfae_c1b_dgrad")
    elif Wt is not None and Wt.numel() > 0 and Wt.dtype == ops.BF16:
        da = ops.c1b_fwd(Wt, dt, label="wfae_c1b_dgrad")
    else:
        da = ops.conv1x1_bwd_data(dt, w)
    return ops.bn_act_bwd(da, x, gamma, st, dgamma, dbeta, res, 1, training)


class BottleneckFn(Function):
    @staticmethod
    def forward(ctx, x, g1, b1, w1, g2, b2, wg, g3, b3, w3, mod, x_stats=None):
        """`x_stats`: the BatchNorm sums of x if the kernel that produced x already reduced them (the previous
        Bottleneck's last 1x1 convolution); `mod._out_stats` receives those of y when `mod.emit_stats` is set.  In
        training mode the statistics of the two 1x1 outputs (t1 here, y for the next block) ride in the GEMM
        epilogues (fp64 partial sums, STAT_FUSION) instead of costing a pass over the tensor each; the first and — for
        C <= 128 — the third BatchNorm + GELU are applied inside the loaders of the GEMMs that consume them
        (FUSE_A1 / FUSE_A3), so a1 / a3 are neither written nor saved."""
        xc = _c(x)
        if xc is not x:
            x_stats = None
        x = xc
        bn1, bn2, bn3 = mod.f[0], mod.f[3], mod.f[6]
        groups = mod.f[5].groups
        training = _use_batch_stats(bn1)
        fuse = training and STAT_FUSION
        emit = fuse and getattr(mod, "emit_stats", False)
        C, mid, hw = x.shape[1], w1.shape[0], x.shape[2] * x.shape[3]
        # bf16 storage: csrc/c1b.hip serves the four 1x1 products — (M = mid, K = C): the C -> C/4 forward and the C/4 -> C data
        # gradient; (M = C, K = mid): the C/4 -> C forward and the C -> C/4 data gradient — with one bf16 weight plane each way,
        # written once here and reused in backward
        W1p = W3p = (None, None)
        bf = x.dtype == ops.BF16
        rb1, rb3 = bf and ops.c1rb_supported(mid, C, hw), bf and ops.c1rb_supported(C, mid, hw)   # register-direct (csrc/c1rb.hip)
        cb1 = rb1 or (bf and ops.c1b_supported(mid, C, hw))
        cb3 = rb3 or (bf and ops.c1b_supported(C, mid, hw))
        if C >= 256:
            rb1 = False        # the C -> C/4 forward with the prologue runs on c1b there (ops.c1rb_take), so its plane pair is needed
        if (cb1 and not rb1) or (cb3 and not rb3):      # c1b.hip takes prepared bf16 planes; c1rb.hip reads the fp32 weight itself
            W1p, W3p = ops.c1b_weights(w1), ops.c1b_weights(w3)
        st1 = _bn_stats_rows(x_stats if (fuse or isinstance(x_stats, ops.StatParts)) else None, x, bn1, training)
        fused1 = FUSE_A1 and x.shape[1] <= FUSE_A1_MAXC and ops.conv1x1_bnact_supported(x, w1.shape[0])
        a1 = None if fused1 else ops.bn_act_fwd(x, st1, 1)
        if cb1:
            src, pro = (x, st1) if fused1 else (a1, None)
            t1, sr2 = _b16(w1, W1p[0], False, src, pro, None, True) if fuse else (_b16(w1, W1p[0], False, src, pro), None)
        elif fused1:
            t1, sr2 = ops.conv1x1_fwd_bnact(x, st1, w1, stats=True) if fuse else (ops.conv1x1_fwd_bnact(x, st1, w1), None)
        else:
            t1, sr2 = ops.conv1x1_fwd_stats(a1, w1) if fuse else (ops.conv1x1_fwd(a1, w1), None)
        st2 = _bn_stats_rows(sr2, t1, bn2, training)
        a2 = ops.bn_act_fwd(t1, st2, 1)
        t2 = _g3_fwd(a2, wg, groups)
        st3 = _bn_stats(t2, bn3, training)
        reg_direct = ops.c1r_supported(C, mid, hw) if x.dtype == torch.float32 else (bf and ops.c1rb_supported(C, mid, hw))
        a3_maxc = FUSE_A3_MAXC_C1R if reg_direct else FUSE_A3_MAXC
        if FUSE_A3 and w3.shape[0] <= a3_maxc and ops.conv1x1_bnact_supported(t2, w3.shape[0]):
            a3 = None
            if cb3:
                y, mod._out_stats = _b16(w3, W3p[0], False, t2, st3, x, True) if emit else (_b16(w3, W3p[0], False, t2, st3, x), None)
            else:
                y, mod._out_stats = ops.conv1x1_fwd_bnact(t2, st3, w3, None, x, stats=True) if emit else \
                    (ops.conv1x1_fwd_bnact(t2, st3, w3, None, x), None)
        else:
            a3 = ops.bn_act_fwd(t2, st3, 1)
            if cb3:
                y, mod._out_stats = _b16(w3, W3p[0], False, a3, None, x, True) if emit else (_b16(w3, W3p[0], False, a3, None, x), None)
            else:
                y, mod._out_stats = ops.conv1x1_fwd_stats(a3, w3, None, x) if emit else (ops.conv1x1_fwd(a3, w3, None, x), None)
        ctx.save_for_backward(x, _opt(a1), t1, a2, t2, _opt(a3), g1, w1, g2, wg, g3, w3,
                              st1.mean, st1.invstd, st1.scale, st1.shift,
                              st2.mean, st2.invstd, st2.scale, st2.shift,
                              st3.mean, st3.invstd, st3.scale, st3.shift,
                              _opt(W1p[1] if (cb3 and not rb3) else None), _opt(W3p[1] if (cb1 and not rb1) else None))
        ctx.training = training
        ctx.groups = groups
        ctx.betas = (b1, b2, b3)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x, a1, t1, a2, t2, a3, g1, w1, g2, wg, g3, w3, *s) = ctx.saved_tensors
        st1, st2, st3 = _mk_stats(*s[0:4]), _mk_stats(*s[4:8]), _mk_stats(*s[8:12])
        W1t, W3t = s[12], s[13]     # bf16 planes of w1^T (M = C, K = mid) and w3^T (M = mid, K = C); empty: fp32 storage
        tr = ctx.training
        dy = _c(dy)
        mid = w1.shape[0]
        dw3 = grad_buffer(w3)
        if a3.numel() == 0:     # a3 = gelu(bn3(t2)) was never materialised: rebuilt in the weight gradient's loader
            _wgrad(lambda: ops.conv1x1_bwd_weight_bnact(dy, t2, st3, dw3), dy, t2, st3.scale, st3.shift)
        else:
            _wgrad(lambda: ops.conv1x1_bwd_weight(dy, a3, dw3), dy, a3)
        dg3, db3 = grad_buffer(g3), grad_buffer(ctx.betas[2])
        dt2 = _dgrad_bn(dy, w3, W3t, t2, g3, st3, dg3, db3, None, tr)
        dwg = grad_buffer(wg)
        _wgrad(lambda: _g3_wgrad(dt2, a2, dwg, ctx.groups), dt2, a2)
        da2 = _g3_dgrad(dt2, wg, mid, ctx.groups)
        del dt2
        dg2, db2 = grad_buffer(g2), grad_buffer(ctx.betas[1])
        dt1 = ops.bn_act_bwd(da2, t1, g2, st2, dg2, db2, None, 1, tr)
        del da2
        dw1 = grad_buffer(w1)
        if a1.numel() == 0:     # a1 was never materialised: the weight gradient rebuilds it from x in its loader
            _wgrad(lambda: ops.conv1x1_bwd_weight_bnact(dt1, x, st1, dw1), dt1, x, st1.scale, st1.shift)
        else:
            _wgrad(lambda: ops.conv1x1_bwd_weight(dt1, a1, dw1), dt1, a1)
        dg1, db1 = grad_buffer(g1), grad_buffer(ctx.betas[0])
        dx = _dgrad_bn(dt1, w1, W1t, x, g1, st1, dg1, db1, dy, tr)
        return dx, dg1, db1, dw1, dg2, db2, dwg, dg3, db3, dw3, None, None


# ------------------------------------------------------------ leaf layers --
class Conv1x1Fn(Function):
    """y = conv1x1(x, w) + bias (+ pos broadcast over the batch)."""

    @staticmethod
    def forward(ctx, x, w, bias, pos):
        # the latent projections (enc[4], dec[0]) are fp32 layers: in the bf16-storage mode the (small) neighbouring
        # activation is converted at this boundary, the result is fp32
        ctx.x_dtype = x.dtype
        x = ops.to_f32(_c(x))
        y = ops.conv1x1_fwd(x, w, bias, None if pos is None else _c(pos), pos is not None)
        ctx.save_for_backward(x, w)
        ctx.bias, ctx.pos = bias, pos
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = ops.to_f32(_c(dy))
        nb, cout, h, wd = dy.shape
        dw = grad_buffer(w)
        ops.conv1x1_bwd_weight(dy, x, dw)
        dx = ops.to_dtype(ops.conv1x1_bwd_data(dy, w), ctx.x_dtype) if ctx.needs_input_grad[0] else None
        dbias = dpos = None
        if ctx.bias is not None:
            dbias = grad_buffer(ctx.bias)
            ops.reduce_sum(dy, nb, cout, h * wd, dbias)
        if ctx.pos is not None:
            dpos = grad_buffer(ctx.pos)
            ops.reduce_sum(dy, nb, cout * h * wd, 1, dpos)
        return dx, dw, dbias, dpos


def conv1x1(x, w, bias=None, pos=None):
    return Conv1x1Fn.apply(x, w, bias, pos)


class LinearFn(Function):
    @staticmethod
    def forward(ctx, x, w, bias):
        x = _c(x)
        y = ops.linear_fwd(x, w, bias)
        ctx.save_for_backward(x, w)
        ctx.bias = bias
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = _c(dy)
        dw = grad_buffer(w)
        ops.linear_bwd_weight(dy, x, dw)
        dx = ops.linear_bwd_data(dy, w) if ctx.needs_input_grad[0] else None
        dbias = None
        if ctx.bias is not None:
            dbias = grad_buffer(ctx.bias)
            ops.reduce_sum(dy, dy.shape[0], w.shape[0], 1, dbias)
        return dx, dw, dbias


class DConvFn(Function):
    """3x3 'same' stride-1 convolution with any group count (direct kernels)."""

    @staticmethod
    def forward(ctx, x, w, bias, groups):
        x = _c(x)
        y = ops.dconv_fwd(x, w, bias, 3, 1, 1, groups)
        ctx.save_for_backward(x, w)
        ctx.groups = groups
        ctx.bias = bias
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = _c(dy)
        dw = grad_buffer(w)
        ops.dconv_bwd_weight(dy, x, dw, 3, 1, 1, ctx.groups)
        dx = ops.dconv_bwd_data(dy, w, x.shape[1], 3, 1, ctx.groups, out_dtype=x.dtype) if ctx.needs_input_grad[0] else None
        dbias = None
        if ctx.bias is not None:
            nb, cout, h, wd = dy.shape
            dbias = grad_buffer(ctx.bias)
            ops.reduce_sum(dy, nb, cout, h * wd, dbias)
        return dx, dw, dbias, None


class Conv4x4DownFn(Function):
    @staticmethod
    def forward(ctx, x, w):
        x = _c(x)
        ctx.save_for_backward(x, w)
        return _down_fwd(x, w)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = _c(dy)
        dw = grad_buffer(w)
        _down_wgrad(dy, x, dw)
        dx = ops.conv4x4s2_up(dy, w) if ctx.needs_input_grad[0] else None
        return dx, dw


class ConvT4x4UpFn(Function):
    @staticmethod
    def forward(ctx, x, w):
        x = _c(x)
        ctx.save_for_backward(x, w)
        return ops.conv4x4s2_up(x, w)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = _c(dy)
        dw = grad_buffer(w)
        ops.conv4x4s2_wgrad(x, dy, dw)
        dx = ops.conv4x4s2_down(dy, w) if ctx.needs_input_grad[0] else None
        return dx, dw


class BatchNormActFn(Function):
    """BatchNorm2d (+ optional fused GELU) as a standalone layer."""

    @staticmethod
    def forward(ctx, x, gamma, beta, bn, act):
        x = _c(x)
        training = _use_batch_stats(bn)
        st = _bn_stats(x, bn, training)
        y = ops.bn_act_fwd(x, st, act)
        ctx.save_for_backward(x, gamma, st.mean, st.invstd, st.scale, st.shift)
        ctx.training, ctx.act, ctx.beta = training, act, beta
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, *s = ctx.saved_tensors
        st = _mk_stats(*s)
        dg, db = grad_buffer(gamma), grad_buffer(ctx.beta)
        dx = ops.bn_act_bwd(_c(dy), x, gamma, st, dg, db, None, ctx.act, ctx.training,
                            need_dx=ctx.needs_input_grad[0])
        return dx, dg, db, None, None


class GeluFn(Function):
    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        ctx.save_for_backward(x)
        return ops.gelu_fwd(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return ops.gelu_bwd(_c(dy), x)


class SigmoidFn(Function):
    @staticmethod
    def forward(ctx, x):
        y = ops.sigmoid_fwd(_c(x))
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        return ops.sigmoid_bwd(_c(dy), y)


class AddFn(Function):
    @staticmethod
    def forward(ctx, a, b):
        return ops.add(_c(a), _c(b))

    @staticmethod
    def backward(ctx, dy):
        return dy, dy


# ------------------------------------------------------- latent transformer --
class AddLayerNormFn(Function):
    """LayerNorm(x + res): the post-norm residual of nn.TransformerEncoderLayer."""

    @staticmethod
    def forward(ctx, x, res, gamma, beta, eps):
        x, res = _c(x), _c(res)
        y, mean, rstd = ops.layernorm_fwd(x, res, gamma, beta, eps)
        ctx.save_for_backward(x, res, gamma, mean, rstd)
        ctx.beta = beta
        return y

    @staticmethod
    def backward(ctx, dy):
        x, res, gamma, mean, rstd = ctx.saved_tensors
        dg, db = grad_buffer(gamma), grad_buffer(ctx.beta)
        dh = ops.layernorm_bwd(_c(dy), x, res, gamma, mean, rstd, dg, db)
        return dh, dh, dg, db, None


class MhaSeqFirstFn(Function):
    @staticmethod
    def forward(ctx, qkv, s, n, h, p_drop, seed, batch_first=False):
        qkv = _c(qkv)
        d = qkv.shape[1] // (3 * h)
        out, probs = ops.mha_fwd(qkv, s, n, h, d, p_drop, seed, batch_first)
        ctx.save_for_backward(qkv, probs)
        ctx.cfg = (s, n, h, d, p_drop, seed, batch_first)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, probs = ctx.saved_tensors
        return ops.mha_bwd(qkv, probs, _c(dout), *ctx.cfg), None, None, None, None, None, None


class SingleQueryAttnFn(Function):
    """attention of one (projected) query per batch element over L projected key/value tokens
    (GlobalCrossEncode, pipeline/models/ae_vit.py:22-42)"""

    @staticmethod
    def forward(ctx, q, kv, l, h):
        q, kv = _c(q), _c(kv)
        b = q.shape[0]
        d = q.shape[1] // h
        out, probs = ops.sq_attn_fwd(q, kv, b, l, h, d)
        ctx.save_for_backward(q, kv, probs)
        ctx.cfg = (b, l, h, d)
        return out

    @staticmethod
    def backward(ctx, dout):
        q, kv, probs = ctx.saved_tensors
        dq, dkv = ops.sq_attn_bwd(q, kv, probs, _c(dout), *ctx.cfg)
        return dq, dkv, None, None


class AddBcastFn(Function):
    """x (B, ...) + p (...) with p a parameter broadcast over the leading dimension (positional tokens)"""

    @staticmethod
    def forward(ctx, x, p):
        ctx.p = p
        return ops.add_bcast(_c(x), _c(p))

    @staticmethod
    def backward(ctx, dy):
        dy = _c(dy)
        dp = None
        if ctx.needs_input_grad[1]:
            dp = grad_buffer(ctx.p)
            ops.reduce_sum(dy, dy.numel() // dp.numel(), dp.numel(), 1, dp)
        return dy, dp


class ExpandRowsFn(Function):
    """(1, F) parameter -> (B, F) rows (query_vec.expand, dec_queries.expand in AE_ViT_2048.forward): wfae_copy_rows"""

    @staticmethod
    def forward(ctx, p, b):
        ctx.p = p
        return ops.copy_rows(_c(p.detach()).view(-1), b, p.numel(), 0)   # every row = the parameter (src_ld = 0)

    @staticmethod
    def backward(ctx, dy):
        dp = grad_buffer(ctx.p)
        dy = _c(dy)
        ops.reduce_sum(dy, dy.shape[0], dp.numel(), 1, dp)
        return dp, None


class SliceColsFn(Function):
    """columns [a, b) of a 2-D tensor (a copy; the gradient is zero outside the slice)"""

    @staticmethod
    def forward(ctx, x, a, b):
        x = _c(x)
        ctx.cfg = (x.shape, a, b)
        return ops.copy_rows(x, x.shape[0], b - a, x.shape[1], src_off=a)

    @staticmethod
    def backward(ctx, dy):
        shape, a, b = ctx.cfg
        return ops.copy_rows(_c(dy), shape[0], b - a, b - a, dst_ld=shape[1], dst_off=a, zero_fill=True), None, None


class RepeatRowsFn(Function):
    """(B, F) -> (B*l, F), every row repeated l times (one latent-derived token copied to all positions)"""

    @staticmethod
    def forward(ctx, x, l):
        ctx.l = l
        x = _c(x)
        b, f = x.shape
        return ops.copy_rows(x, b * l, f, f, row_div=l)

    @staticmethod
    def backward(ctx, dy):
        l = ctx.l
        dy = _c(dy)
        return ops.sum_mid(dy, dy.shape[0] // l, l, dy.shape[1]), None


class PatchEmbedFn(Function):
    """Conv2d(C, E, kernel_size=P, stride=P) as patch folding + one GEMM (ae_vit.py:99,138-139): returns the
    token rows (B*Hp*Wp, E), i.e. already `.flatten(2).transpose(1, 2)`"""

    @staticmethod
    def forward(ctx, x, w, bias):
        p = w.shape[2]
        rows = ops.patchify(_c(x), p)
        ctx.save_for_backward(rows, w)
        ctx.bias = bias
        return ops.linear_fwd(rows, w.view(w.shape[0], -1), bias)

    @staticmethod
    def backward(ctx, dy):
        rows, w = ctx.saved_tensors
        dy = _c(dy)
        dw = grad_buffer(w)
        ops.linear_bwd_weight(dy, rows, dw.view(w.shape[0], -1))
        dbias = None
        if ctx.bias is not None:
            dbias = grad_buffer(ctx.bias)
            ops.reduce_sum(dy, dy.shape[0], dy.shape[1], 1, dbias)
        return None, dw, dbias


class UnpatchFn(Function):
    """ConvTranspose2d(E, C, kernel_size=P, stride=P) on token rows (B*Hp*Wp, E) -> image (B,C,Hp*P,Wp*P)
    (ae_vit.py:128,165-166): one GEMM + patch unfolding"""

    @staticmethod
    def forward(ctx, tok, w, bias, b, hp, wp):
        tok = _c(tok)
        e, c, p = w.shape[0], w.shape[1], w.shape[2]
        rows = ops.linear_bwd_data(tok, w.view(e, -1))          # (rows, E) x (E, C*P*P)
        ctx.save_for_backward(tok, w)
        ctx.cfg, ctx.bias = (b, c, hp, wp, p), bias
        return ops.unpatchify(rows, bias, b, c, hp, wp, p)

    @staticmethod
    def backward(ctx, dimg):
        tok, w = ctx.saved_tensors
        b, c, hp, wp, p = ctx.cfg
        drows = ops.patchify(_c(dimg), p)
        e = w.shape[0]
        dw = grad_buffer(w)
        ops.linear_bwd_weight(tok, drows, dw.view(e, -1))        # dw[e][j] = sum_rows tok[row][e] drows[row][j]
        dtok = ops.linear_fwd(drows, w.view(e, -1), None) if ctx.needs_input_grad[0] else None
        dbias = None
        if ctx.bias is not None:
            dbias = grad_buffer(ctx.bias)
            ops.reduce_sum(_c(dimg), b, c, hp * p * wp * p, dbias)
        return dtok, dw, dbias, None, None, None


class ReluFn(Function):
    @staticmethod
    def forward(ctx, x):
        y = ops.relu_fwd(_c(x))
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        return ops.relu_bwd(_c(dy), y)


class DropoutFn(Function):
    """y = x * mask / (1-p); the mask is a pure function of (seed, element index), so backward
    applies the same kernel to dy."""

    @staticmethod
    def forward(ctx, x, p_drop, seed):
        ctx.cfg = (p_drop, seed)
        return ops.dropout(_c(x), p_drop, seed)

    @staticmethod
    def backward(ctx, dy):
        return ops.dropout(_c(dy), *ctx.cfg), None, None


_seed_counter = [0]


def next_seed():
    _seed_counter[0] += 1
    return (torch.initial_seed() * 1000003 + _seed_counter[0]) & 0x7FFFFFFFFFFFFFFF


def dropout(x, p, training):
    if not training or p <= 0.0:
        return x
    return DropoutFn.apply(x, float(p), next_seed())


# ------------------------------------------------------------------- losses --
class L1LossFn(Function):
    """weight * mean |recon - x|  (F.l1_loss, experiments/ae_v2/train.py:55)."""

    @staticmethod
    def forward(ctx, recon, x, weight):
        recon, x = _c(recon), _c(x)
        ctx.save_for_backward(recon, x)
        ctx.weight = float(weight)
        return ops.l1_fwd(recon, x, ctx.weight)

    @staticmethod
    def backward(ctx, g):
        recon, x = ctx.saved_tensors
        return ops.l1_bwd(recon, x, _c(g), ctx.weight), None, None


class SsimFn(Function):
    """pytorch_msssim.ssim(X, Y, data_range=1): gradient flows to Y (train.py:62)."""

    @staticmethod
    def forward(ctx, x, y):
        x, y = _c(x), _c(y)
        ctx.save_for_backward(x, y)
        return ops.ssim_fwd(x, y, False)

    @staticmethod
    def backward(ctx, g):
        x, y = ctx.saved_tensors
        return None, ops.ssim_bwd(x, y, _c(g))


class MseLossFn(Function):
    """F.mse_loss(pred, target) (reference v1_experiments/pretrained_ae_linear_sevir/train.py:82)"""

    @staticmethod
    def forward(ctx, pred, target):
        pred, target = _c(pred), _c(target)
        ctx.save_for_backward(pred, target)
        return ops.mse_fwd(pred, target)

    @staticmethod
    def backward(ctx, g):
        pred, target = ctx.saved_tensors
        return ops.mse_bwd(pred, target, _c(g)), None


def mse_loss(pred, target):
    return MseLossFn.apply(pred, target)


def l1_loss(recon, x, weight=1.0):
    return L1LossFn.apply(recon, x, weight)


def ssim(x, y):
    return SsimFn.apply(x, y)
