"""Host-side mirror of the reference's pipeline/helpers.py for the AE train step:
optimiser / scheduler factories with the reference signatures, the YAML
override check and the gradient-norm tracker.  (W&B / Lightning glue of the
reference file is orchestration and out of scope — SURVEY.md §2 row 8.)"""
from __future__ import annotations

import math

import torch

from .. import ops
from ..optim import CosineWarmupLR, FusedAdamW


def adamw_optimizer(model, lr, weight_decay, beta1=0.9, beta2=0.999):
    """reference pipeline/helpers.py:63-74 -> torch.optim.AdamW(model.parameters(), ...);
    here one fused kernel over a flat parameter arena with identical arithmetic."""
    return FusedAdamW(model.parameters(), lr=lr, weight_decay=weight_decay, betas=(beta1, beta2))


def cosine_warmup_scheduler(opt, start_lr, final_lr, peak_lr, total_steps, warmup_steps):
    """reference pipeline/helpers.py:76-107 (same positional order). The optimiser's lr is
    overridden by peak_lr like the reference does (:86-89)."""
    for g in opt.param_groups:
        if g["lr"] != peak_lr:
            print(f"lr is not peak lr, it is {g['lr']} changing to {peak_lr}")
            g["lr"] = peak_lr
    return CosineWarmupLR(opt, start_lr, final_lr, peak_lr, total_steps, warmup_steps)


def one_cycle_scheduler(opt, start_lr, peak_lr, final_lr, total_steps, rampup_steps):
    """reference pipeline/helpers.py:109-140 (same positional order): OneCycleLR with pct_start = rampup/total,
    div_factor = peak/start, final_div_factor = start/final, cosine annealing."""
    from ..optim import OneCycleLR
    if rampup_steps / total_steps < 0.2:
        print(f"rampup steps should be higher than 20% of total steps, it is {rampup_steps / total_steps}")
    return OneCycleLR(opt, max_lr=peak_lr, total_steps=total_steps, pct_start=rampup_steps / total_steps,
                      div_factor=peak_lr / start_lr, final_div_factor=start_lr / final_lr)


def check_yaml(cfg, cli_cfg, path=""):
    """reference pipeline/helpers.py:260-266: reject override keys absent from the base file."""
    for k in cli_cfg:
        full_key = f"{path}.{k}" if path else k
        if k not in cfg:
            raise KeyError(f"Invalid override key: '{full_key}' not found in base config")
        if isinstance(cli_cfg[k], dict) and isinstance(cfg[k], dict):
            check_yaml(cfg[k], cli_cfg[k], full_key)


def grad_norm(optimizer_or_params, norm_type=2):
    """TrackGradNormCallback arithmetic (reference :250-256) without 323 host syncs: one
    sum-of-squares kernel per gradient arena (or per tensor), one .item()."""
    assert norm_type == 2
    from ..functional import join_side_stream
    join_side_stream()
    total = 0.0
    arenas = getattr(optimizer_or_params, "arenas", None)
    if arenas:
        for a in arenas:
            if a.grads_in_arena():
                total += float(ops.sumsq(a.flat_g).item())
            else:
                total += sum(float(ops.sumsq(p.grad.contiguous().view(-1)).item()) for p in a.params if p.grad is not None)
        return math.sqrt(total)
    for p in optimizer_or_params:
        if p.grad is not None:
            total += float(ops.sumsq(p.grad.contiguous().view(-1)).item())
    return math.sqrt(total)


def log_metrics(pred, target, tag, log_fn=None):
    """reference :142-153: calc_metrics -> {tag_key: value}."""
    from .metrics import calc_metrics
    out = {f"{tag}_{k}": v for k, v in calc_metrics(pred, target).items()}
    if log_fn is not None:
        log_fn(out)
    return out
