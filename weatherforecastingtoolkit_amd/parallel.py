"""Data parallelism for the AE train step: one process per GPU, RCCL over xGMI.

The reference has no explicit distributed code; with >1 device Lightning would
wrap the model in DDP (experiments/ae_v2/train.py:332-343):
gradient all-reduce (mean) per step, parameters broadcast from rank 0 at start,
BatchNorm statistics per replica (no SyncBN), buffers broadcast from rank 0.

MI355X design: gradients already live in ONE flat fp32 arena (optim.FlatArena),
so the exchange is a handful of large in-place all-reduces (bucket_mb each —
xGMI is point-to-point, big messages amortise the per-link latency) issued on
the current stream right after backward; the 1/world_size mean is folded into
the AdamW kernel (FusedAdamW.grad_scale), so no extra pass over the gradients.
Works with any torch.distributed backend (nccl = RCCL on ROCm; gloo on CPU for
the world_size-2 tests).
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* if WORLD_SIZE > 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return 0, 1, 0
    rank, local = int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", os.environ["RANK"]))
    if not dist.is_initialized():
        # SURVEY.md 8(e) "Determinism": the sum order of an all-reduce is fixed by the algorithm (ring order and chunking)
        # and by the bucket sequence.  Buckets are fixed-size slices of ONE flat arena walked in address order (GradSync),
        # and the RCCL algorithm is pinned here unless the caller chose one, so the same ranks give the same sums run to run.
        os.environ.setdefault("NCCL_ALGO", "Ring")
        if backend is None:
            # WFAE_DIST_BACKEND=gloo: several ranks sharing one card (rehearsals on a 1-GPU box; RCCL needs a device per rank)
            backend = os.environ.get("WFAE_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


class GradSync:
    """Bucketed in-place sum all-reduce of flat gradient buffers."""

    def __init__(self, bucket_mb=256, group=None):
        self.bucket = max(1, int(bucket_mb)) * (1 << 20) // 4
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1

    def allreduce_(self, flat: torch.Tensor):
        if self.world == 1:
            return flat
        n = flat.numel()
        for o in range(0, n, self.bucket):
            dist.all_reduce(flat[o:min(n, o + self.bucket)], op=dist.ReduceOp.SUM, group=self.group)
        return flat

    def broadcast_(self, flat: torch.Tensor, src=0):
        if self.world > 1:
            dist.broadcast(flat, src=src, group=self.group)
        return flat


class DataParallelTrainer:
    """Glue: replicate parameters from rank 0, all-reduce the gradient arena, let FusedAdamW apply
    the 1/world mean.

    Overlap: parameters sit in the arena in registration order (pos_emb, enc.*, to_latent, from_latent,
    dec.*) and backward finishes them from the end.  Backward hooks on the two big Linear layers —
    150 M of the 215 M parameters at 384x384 — start an asynchronous all-reduce of the arena tail
    [from_latent .. end] and then of to_latent as soon as their gradients exist, so ~85 % of the bytes
    travel over xGMI while the encoder backward (the second half of the step) is still computing; only the
    encoder's 132 MB are exchanged after backward."""

    def __init__(self, model, optimizer, bucket_mb=256, overlap=None):
        self.model, self.opt = model, optimizer
        self.sync = GradSync(bucket_mb)
        self.world = self.sync.world
        optimizer.grad_scale = 1.0 / self.world
        for a in optimizer.arenas:
            self.sync.broadcast_(a.flat_p, 0)
        self.sync_buffers()
        self._pending = []       # [(handle, start, end)]
        self._done_from = None   # arena offset from which gradients are already being reduced
        self._hooks = []
        self._defer = False      # inside no_sync(): backward hooks start no exchange
        self.hook_launches = 0   # asynchronous all-reduces started from backward hooks (tests / diagnostics)
        if overlap is None:
            # opt-in until it has been timed on a real 8-GPU xGMI node (this build only had 1-GPU boxes)
            overlap = os.environ.get("WFAE_DP_OVERLAP", "0") == "1"
        if overlap and self.world > 1 and len(optimizer.arenas) == 1:
            self._install_hooks()

    # -- overlap ---------------------------------------------------------------------------------
    def _offset_of(self, param):
        a = self.opt.arenas[0]
        for p, o in zip(a.params, a.offsets):
            if p is param:
                return o
        return None

    def _install_hooks(self):
        for name in ("from_latent", "to_latent"):
            mod = getattr(self.model, name, None)
            if mod is None or not hasattr(mod, "weight"):
                continue
            off = self._offset_of(mod.weight)
            if off is None:
                continue
            self._hooks.append(mod.register_full_backward_hook(self._make_hook(off)))

    def _make_hook(self, start):
        def hook(module, grad_input, grad_output):
            if self._defer:
                return
            if self._done_from is not None and start >= self._done_from:
                # a second backward before reduce_gradients() would ADD un-reduced gradients to slots whose
                # exchange is already in flight: the ranks would silently diverge
                raise RuntimeError("DataParallelTrainer: backward ran twice before reduce_gradients(); wrap the "
                                   "micro-batches of a gradient-accumulation step except the last in dp.no_sync()")
            self._launch_tail(start)
        return hook

    def no_sync(self):
        """context for the micro-batches of a gradient-accumulation step EXCEPT the last: their backward passes
        only accumulate into the arena, the hook-driven exchange starts with the last micro-batch (DDP.no_sync)"""
        dp = self

        class _NoSync:
            def __enter__(self):
                self.prev, dp._defer = dp._defer, True

            def __exit__(self, *exc):
                dp._defer = self.prev

        return _NoSync()

    def _launch_tail(self, start):
        """all-reduce arena[start : previously launched start) asynchronously"""
        from .functional import join_side_stream
        a = self.opt.arenas[0]
        end = a.numel if self._done_from is None else self._done_from
        if start >= end:
            return
        join_side_stream()  # weight gradients of the finished layers may still be on the side stream
        n = self.sync.bucket
        for o in range(start, end, n):
            h = dist.all_reduce(a.flat_g[o:min(end, o + n)], op=dist.ReduceOp.SUM, group=self.sync.group, async_op=True)
            self._pending.append(h)
            self.hook_launches += 1
        self._done_from = start

    def sync_buffers(self, module=None):
        """rank 0's BatchNorm running statistics to every rank (DDP broadcast_buffers parity);
        call before evaluation / checkpointing.  `module`: the module whose buffers travel (default: the wrapped
        model; the train scripts pass the whole LightningModule counterpart so that loss.discriminator's BatchNorm
        buffers, which are saved from rank 0, are rank 0's everywhere)."""
        if self.world == 1:
            return
        for b in (self.model if module is None else module).buffers():
            if b.dtype.is_floating_point:
                dist.broadcast(b, src=0)

    def wait_pending(self):
        """block until the hook-started all-reduces have landed (reduce_gradients does this itself)"""
        for h in self._pending:
            h.wait()
        self._pending.clear()

    def unused_slots_are_zero(self):
        """True when every arena slot of a parameter WITHOUT a gradient (the never-used `tf_encoder.*` template of
        the _tf model) holds exact zeros — those slots ride along in the all-reduce, so they must be
        the same on every rank (what DDP's find_unused_parameters guarantees by other means)"""
        for a in self.opt.arenas:
            for p, o in zip(a.params, a.offsets):
                if p.grad is None and bool(a.flat_g[o:o + p.numel()].any()):
                    return False
        return True

    def start_reduce(self):
        """Begin the exchange of this optimiser's gradients WITHOUT waiting for it (call after its backward): every
        bucket goes out as an asynchronous all-reduce, ordered behind everything queued on the compute stream so far and
        running on the process group's own stream, so whatever is queued AFTER this call overlaps the exchange — the
        discriminator's forward / backward in the AE+GAN step (reference experiments/ae_v2_2/train.py:126-159: its loss
        reads the detached reconstruction and the discriminator's own parameters only).  finish_reduce() must follow
        before the gradients are read or the optimiser steps.  Same buckets, same order, same sums as reduce_gradients()."""
        from .functional import join_side_stream
        join_side_stream()
        self._started = True
        if self.world == 1:
            return
        for a in self.opt.arenas:
            # gradients that autograd SUMMED from several uses of a parameter (the discriminator on the real and the
            # fake batch, ae_v2_2/train.py:88-89) live in fresh buffers outside the arena (functional.grad_buffer):
            # exchange those tensor by tensor — FusedAdamW consumes them from p.grad as well.  The graph is the same
            # on every rank, so every rank issues the same sequence of collectives.
            runs, stray = a.runs()
            for i in stray:
                g = a.params[i].grad
                if not g.is_contiguous():
                    raise RuntimeError("data-parallel step: non-contiguous gradient outside the arena")
                self._pending.append(dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.sync.group, async_op=True))
        n = self.sync.bucket
        for k, a in enumerate(self.opt.arenas):
            # parameters without a gradient (never-used template layers) contribute their zero-initialised
            # arena slots: the same on every rank, like DDP with find_unused_parameters
            end = self._done_from if (k == 0 and self._done_from is not None) else a.numel
            for o in range(0, end, n):
                self._pending.append(dist.all_reduce(a.flat_g[o:min(end, o + n)], op=dist.ReduceOp.SUM, group=self.sync.group,
                                                     async_op=True))
        self._done_from = None

    def finish_reduce(self):
        """wait (the compute stream, for NCCL / RCCL; the host, for gloo) until the exchange start_reduce() began has landed"""
        if not getattr(self, "_started", False):
            raise RuntimeError("DataParallelTrainer.finish_reduce() without start_reduce()")
        self._started = False
        self.wait_pending()

    def reduce_gradients(self):
        """call after loss.backward(): exchanges whatever the backward hooks have not started yet and
        waits for the asynchronous parts"""
        self.start_reduce()
        self.finish_reduce()
