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
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
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
    """Glue: replicate parameters from rank 0, all-reduce the gradient arena after
    backward, let FusedAdamW apply the 1/world mean."""

    def __init__(self, model, optimizer, bucket_mb=256):
        self.model, self.opt = model, optimizer
        self.sync = GradSync(bucket_mb)
        self.world = self.sync.world
        optimizer.grad_scale = 1.0 / self.world
        for a in optimizer.arenas:
            self.sync.broadcast_(a.flat_p, 0)
        self.sync_buffers()

    def sync_buffers(self):
        """rank 0's BatchNorm running statistics to every rank (DDP broadcast_buffers parity);
        call before evaluation / checkpointing."""
        if self.world == 1:
            return
        for b in self.model.buffers():
            if b.dtype.is_floating_point:
                dist.broadcast(b, src=0)

    def reduce_gradients(self):
        from .functional import join_side_stream
        join_side_stream()
        for a in self.opt.arenas:
            runs, stray = a.runs()
            if stray:
                raise RuntimeError("data-parallel step needs every gradient inside the flat arena")
            # parameters without a gradient (never-used template layers) contribute their zero-initialised
            # arena slots: the same on every rank, like DDP with find_unused_parameters
            self.sync.allreduce_(a.flat_g)
