"""Fused AdamW over flat parameter / gradient / moment arenas, and the
LinearLR -> CosineAnnealingLR schedule of the reference in closed form.

Reference semantics: torch.optim.AdamW built by adamw_optimizer
(pipeline/helpers.py:63-74) and SequentialLR[LinearLR, CosineAnnealingLR]
built by cosine_warmup_scheduler (pipeline/helpers.py:76-107), stepped once
per optimiser step (experiments/ae_v2/train.py:254-261).

MI355X design: all parameters of a group live in ONE contiguous fp32 buffer
(parameters become views), gradients are produced by the backward kernels
directly inside a second contiguous buffer (functional.grad_buffer), so the
optimiser step is one kernel launch and the data-parallel gradient exchange is
an RCCL all-reduce on one buffer (parallel.py).
"""
from __future__ import annotations

import bisect
import math

import torch

from . import ops


class FlatArena:
    """Contiguous storage for a list of parameters, their grads and AdamW moments."""

    def __init__(self, params):
        self.params = [p for p in params]
        dev = self.params[0].device
        self.offsets = []
        off = 0
        for p in self.params:
            self.offsets.append(off)
            off += (p.numel() + 3) // 4 * 4  # keep every view 16-byte aligned
        self.numel = off
        self.flat_p = torch.zeros(off, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(off, dtype=torch.float32, device=dev)
        for p, o in zip(self.params, self.offsets):
            n = p.numel()
            self.flat_p[o:o + n].view_as(p).copy_(p.data)
            p.data = self.flat_p[o:o + n].view_as(p)
            p._wfae_grad_view = self.flat_g[o:o + n].view_as(p)

    def grads_in_arena(self):
        """every parameter has a gradient and it lives inside the arena"""
        for p in self.params:
            if p.grad is None or p.grad.data_ptr() != p._wfae_grad_view.data_ptr():
                return False
        return True

    def runs(self):
        """Maximal runs [(start, end)] (element offsets) of consecutive parameters whose gradient lives in
        the arena, plus the list of parameter indices that have a gradient elsewhere.  Parameters without
        a gradient (e.g. the never-used `tf_encoder.*` template of the _tf model) are skipped, like
        torch.optim does."""
        runs, stray, start, end = [], [], None, None
        for i, p in enumerate(self.params):
            o, n = self.offsets[i], p.numel()
            inside = p.grad is not None and p.grad.data_ptr() == p._wfae_grad_view.data_ptr()
            if inside:
                if start is None:
                    start = o
                end = o + n
            else:
                if start is not None:
                    runs.append((start, end))
                    start = None
                if p.grad is not None:
                    stray.append(i)
        if start is not None:
            runs.append((start, end))
        return runs, stray


class FusedAdamW(torch.optim.Optimizer):
    """torch.optim.AdamW semantics (eps 1e-8, decoupled weight decay, no amsgrad),
    one wfae_adamw launch per parameter group when all gradients sit in the arena."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, flatten=True):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self.grad_scale = 1.0  # set to 1/world_size by the data-parallel wrapper
        self._arenas = []
        for g in self.param_groups:
            ps = [p for p in g["params"] if p.requires_grad]
            g["step"] = 0
            if flatten and ps and ps[0].is_cuda:
                arena = FlatArena(ps)
                arena.m = torch.zeros_like(arena.flat_p)
                arena.v = torch.zeros_like(arena.flat_p)
            else:
                arena = None
            self._arenas.append(arena)

    @property
    def arenas(self):
        return [a for a in self._arenas if a is not None]

    # -- checkpointing ---------------------------------------------------------------------------
    # The moments live in the arenas, outside torch's per-parameter `self.state`, so the inherited state_dict() would
    # silently drop them.  Lightning restores optimiser + scheduler state on `--resume` (reference
    # experiments/ae_v2/train.py:322,346 -> trainer.fit(ckpt_path=...)): a resumed run continues the continuous one.
    def state_dict(self):
        groups, state = [], {}
        for gi, (g, arena) in enumerate(zip(self.param_groups, self._arenas)):
            groups.append({k: v for k, v in g.items() if k != "params"})
            if arena is not None:
                state[gi] = {"layout": [int(p.numel()) for p in arena.params],
                             "exp_avg": arena.m.detach().cpu().clone(), "exp_avg_sq": arena.v.detach().cpu().clone()}
            else:
                state[gi] = {"layout": None,
                             "per_param": [({"m": self.state[p]["m"].cpu().clone(), "v": self.state[p]["v"].cpu().clone()}
                                            if self.state.get(p) else None) for p in g["params"]]}
        return {"state": state, "param_groups": groups, "grad_scale": self.grad_scale, "format": "wfae.FusedAdamW/1"}

    def load_state_dict(self, sd):
        if sd.get("format") != "wfae.FusedAdamW/1":
            raise ValueError("FusedAdamW.load_state_dict: not a FusedAdamW state (format tag missing)")
        if len(sd["param_groups"]) != len(self.param_groups):
            raise ValueError("FusedAdamW.load_state_dict: parameter-group count differs")
        for gi, (g, arena) in enumerate(zip(self.param_groups, self._arenas)):
            g.update({k: (tuple(v) if k == "betas" else v) for k, v in sd["param_groups"][gi].items()})
            st = sd["state"][gi]
            if arena is not None:
                if st["layout"] != [int(p.numel()) for p in arena.params]:
                    raise ValueError("FusedAdamW.load_state_dict: parameter layout of the arena differs")
                arena.m.copy_(st["exp_avg"].to(arena.m.device))
                arena.v.copy_(st["exp_avg_sq"].to(arena.v.device))
            else:
                for p, ps in zip(g["params"], st["per_param"]):
                    if ps is not None:
                        self.state[p] = {"m": ps["m"].to(p.device), "v": ps["v"].to(p.device)}

    def _grad_chunks(self):
        """flat views covering every gradient this optimiser would consume"""
        chunks = []
        for g, arena in zip(self.param_groups, self._arenas):
            if arena is not None:
                runs, stray = arena.runs()
                chunks += [arena.flat_g[a:b] for a, b in runs]
                chunks += [arena.params[i].grad.view(-1) for i in stray if arena.params[i].grad.is_contiguous()]
                if any(not arena.params[i].grad.is_contiguous() for i in stray):
                    raise RuntimeError("clip_grad_norm_: non-contiguous gradient outside the arena")
            else:
                chunks += [p.grad.view(-1) for p in g["params"] if p.grad is not None]
        return chunks

    @torch.no_grad()
    def clip_grad_norm_(self, max_norm):
        """torch.nn.utils.clip_grad_norm_(params, max_norm) (what Lightning's clip_gradients calls,
        experiments/ae_v2_2/train.py:140,155) over the flat gradient arenas, without a host sync:
        sum-of-squares per contiguous run -> device-side coefficient -> in-place scale.  Returns the
        total norm as a 0-dim device tensor."""
        from .functional import join_side_stream
        join_side_stream()
        chunks = self._grad_chunks()
        if not chunks:
            return torch.zeros((), device="cuda")
        parts = torch.empty(len(chunks), dtype=torch.float64, device=chunks[0].device)
        for i, c in enumerate(chunks):
            ops.sumsq_into(c, parts[i])
        coef = ops.clip_coef(parts, max_norm, self.grad_scale)
        for c in chunks:
            ops.scale(c, 1.0, coef[0:1], out=c)
        return coef[1]

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        from .functional import join_side_stream
        join_side_stream()  # weight gradients may have been produced on the side stream
        for g, arena in zip(self.param_groups, self._arenas):
            g["step"] += 1
            t = g["step"]
            b1, b2 = g["betas"]
            bc1, bc2 = 1.0 - b1 ** t, 1.0 - b2 ** t
            args = (g["lr"], b1, b2, g["eps"], g["weight_decay"], bc1, bc2, self.grad_scale)
            if arena is not None:
                runs, stray = arena.runs()
                for a, b in runs:  # one launch per contiguous run (one run when every parameter has a grad)
                    ops.adamw_(arena.flat_p[a:b], arena.flat_g[a:b], arena.m[a:b], arena.v[a:b], *args)
                for i in stray:
                    p = arena.params[i]
                    o, n = arena.offsets[i], p.numel()
                    ops.adamw_(arena.flat_p[o:o + n], p.grad.contiguous().view(-1), arena.m[o:o + n],
                               arena.v[o:o + n], *args)
                continue
            for i, p in enumerate(g["params"]):
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st["m"], st["v"] = torch.zeros_like(p).view(-1), torch.zeros_like(p).view(-1)
                ops.adamw_(p.data.view(-1), p.grad.contiguous().view(-1), st["m"], st["v"], *args)
        return loss


# ------------------------------------------------------------------ schedule
class CosineWarmupLR:
    """Closed form of SequentialLR([LinearLR(start_factor=start/peak, total_iters=warmup),
    CosineAnnealingLR(T_max=total-warmup, eta_min=final)], milestones=[warmup]).

    Reproduces torch's behaviour for NON-integer `warmup_steps` too
    (experiments/ae_v2/train.py:258 passes warmup_ratio*total_steps unrounded):
    the cosine phase is then entered through torch's chained form starting from
    whatever the linear ramp reached at the last integer step below the milestone."""

    def __init__(self, opt, start_lr, final_lr, peak_lr, total_steps, warmup_steps):
        self.opt = opt
        self.start_lr, self.final_lr, self.peak_lr = float(start_lr), float(final_lr), float(peak_lr)
        self.total_steps, self.warmup_steps = total_steps, warmup_steps
        self.last_epoch = 0
        for g in opt.param_groups:
            g["initial_lr"] = self.peak_lr
        self._apply()

    def lr_at(self, e):
        return cosine_warmup_lr(e, self.start_lr, self.peak_lr, self.final_lr, self.total_steps, self.warmup_steps)

    def _apply(self):
        lr = self.lr_at(self.last_epoch)
        for g in self.opt.param_groups:
            g["lr"] = lr
        self._last_lr = [lr for _ in self.opt.param_groups]

    def step(self):
        self.last_epoch += 1
        self._apply()

    def get_last_lr(self):
        return self._last_lr

    def state_dict(self):
        return {"last_epoch": self.last_epoch}

    def load_state_dict(self, sd):
        self.last_epoch = sd["last_epoch"]
        self._apply()


def cosine_warmup_lr(e, start_lr, peak_lr, final_lr, total_steps, warmup_steps):
    sf = start_lr / peak_lr

    def linear(k):
        if warmup_steps <= 0:
            return peak_lr
        return peak_lr * (sf + (1.0 - sf) * min(k, warmup_steps) / warmup_steps)

    if bisect.bisect_right([warmup_steps], e) == 0:
        return linear(e)
    t_max = total_steps - warmup_steps
    if float(warmup_steps).is_integer():
        # milestone hit exactly: SequentialLR calls cosine.step(0) -> closed form from peak_lr
        t = e - int(warmup_steps)
        return final_lr + (peak_lr - final_lr) * (1.0 + math.cos(math.pi * t / t_max)) / 2.0
    # milestone skipped: the cosine scheduler is entered through its chained
    # (recursive) form with last_epoch = 0 at e = floor(warmup)+1, starting from the
    # LR the linear ramp left behind; the product of its per-step ratios telescopes to
    #   (1 + cos(pi t / T)) / (1 + cos(pi (-1) / T)).
    first = math.floor(warmup_steps) + 1
    t, lr0 = e - first, linear(first - 1)
    return final_lr + (lr0 - final_lr) * (1.0 + math.cos(math.pi * t / t_max)) / (1.0 + math.cos(math.pi / t_max))


class OneCycleLR:
    """Closed form of torch.optim.lr_scheduler.OneCycleLR(max_lr, total_steps, pct_start, div_factor,
    final_div_factor, anneal_strategy='cos', three_phase=False, cycle_momentum=True) as the reference's
    one_cycle_scheduler builds it (pipeline/helpers.py:109-140): lr anneals initial -> max over the first
    pct_start*total - 1 steps and max -> initial/final_div_factor over the rest; with Adam-type optimisers beta1
    is cycled 0.95 -> 0.85 -> 0.95 alongside (torch's default base_momentum / max_momentum)."""

    def __init__(self, opt, max_lr, total_steps, pct_start, div_factor, final_div_factor, base_momentum=0.85,
                 max_momentum=0.95, cycle_momentum=True):
        self.opt, self.total_steps = opt, int(total_steps)
        self.max_lr = float(max_lr)
        self.initial_lr = self.max_lr / float(div_factor)
        self.min_lr = self.initial_lr / float(final_div_factor)
        self.end1 = float(pct_start * self.total_steps) - 1.0
        self.end2 = float(self.total_steps - 1)
        self.base_m, self.max_m, self.cycle_momentum = float(base_momentum), float(max_momentum), cycle_momentum
        self.last_epoch = 0
        self._apply()

    @staticmethod
    def _cos(start, end, pct):
        return end + (start - end) / 2.0 * (math.cos(math.pi * pct) + 1.0)

    def values_at(self, step):
        if step > self.total_steps:
            raise ValueError(f"Tried to step {step} times. The specified number of total steps is {self.total_steps}")
        if step <= self.end1 or self.end1 >= self.end2:
            pct = step / self.end1 if self.end1 > 0 else 1.0
            return self._cos(self.initial_lr, self.max_lr, pct), self._cos(self.max_m, self.base_m, pct)
        pct = (step - self.end1) / (self.end2 - self.end1)
        return self._cos(self.max_lr, self.min_lr, pct), self._cos(self.base_m, self.max_m, pct)

    def _apply(self):
        lr, mom = self.values_at(self.last_epoch)
        for g in self.opt.param_groups:
            g["lr"] = lr
            if self.cycle_momentum and "betas" in g:
                g["betas"] = (mom, g["betas"][1])
        self._last_lr = [lr for _ in self.opt.param_groups]

    def step(self):
        self.last_epoch += 1
        self._apply()

    def get_last_lr(self):
        return self._last_lr

    def state_dict(self):
        return {"last_epoch": self.last_epoch}

    def load_state_dict(self, sd):
        self.last_epoch = sd["last_epoch"]
        self._apply()
