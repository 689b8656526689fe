"""experiments/ae_v2_2 on MI355X: the `_lin` autoencoder trained as AE + PatchGAN with manual
optimisation — same CLI, YAML surface, loss arithmetic and step order as the reference's
experiments/ae_v2_2/train.py, without Lightning / W&B.

    python -m weatherforecastingtoolkit_amd.experiments.ae_v2_2.train [--resume True] key=value ...

Step order (reference training_step :126-159):
  pred = AE(inp)                                                        (one forward, :131)
  G: Loss(optimizer_idx=0) = L1 [+ d_weight * -mean(D(pred)) once global_step >= disc_start]
     -> backward (discriminator frozen) -> clip-by-norm 1.0 -> AdamW -> cosine-warmup step
  D (global_step >= disc_start): hinge(D(inp), D(pred.detach())) -> backward -> clip -> AdamW -> step
LPIPS (perceptual_weight > 0) needs VGG weights from the network and is not built; the shipped config
sets perceptual_weight 0.0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch
import torch.nn as tnn

from ... import config as C
from ... import functional as Fn
from ... import ops
from ... import parallel, synth
from ..._lib import WfaeError
from ...nn import flush_bn_counters
from ...pipeline import helpers
from ...pipeline.datasets.sevire.sevir import SEVIRFrameLoader
from ...pipeline.models.ae_64x8x8_lin import PosAwareAE_TF
from .._gan import GanLoss, frozen

HERE = os.path.dirname(os.path.abspath(__file__))

# Keys of the reference's experiments/ae_v2_2/config.yaml that this build accepts but never reads.
CARRIED_KEYS = {
    "project_name": "ae_v3",
    "lpips": {"disc_beta1": 0.5, "disc_beta2": 0.9, "disc_start_lr": 5e-6, "disc_peak_lr": 5e-5, "disc_final_lr": 5e-7,
              "disc_warmup_ratio": 0.1, "kl_weight": 0.0, "logvar_init": 0.0},
    "dataset": {"num_workers": 8, "aug_mode": 1, "input_frames": 0, "pred_frames": 1, "image_width": 128,
                "image_height": 128, "channels": 1},
    "optim": {"beta1": 0.5, "beta2": 0.9},
    "one_cycle": {"peak_lr": 1e-3, "start_lr": 4e-5, "final_lr": 4e-7, "rampup_ratio": 0.3},
    "lr_range_test": {"max_lr": 1, "num_iter": 100},
    "trainer": {"devices": [0], "total_val_steps": -1, "total_test_steps": -1, "save_every_n_steps": 0.1,
                "save_on_train_epoch_end": False, "limit_val_batches": 0.01, "limit_test_batches": None},
    "logging": {"wandb_watch_log_freq": 0, "log_train_all_metrics_n": 0.01, "log_train_plots_n": 0.01, "log_val_plots_n": 0.01},
}



class Loss(GanLoss):
    """reference experiments/ae_v2_2/train.py:29-95"""

    def __init__(self, disc_start, disc_num_layers=3, disc_in_channels=1, disc_weight=1.0, use_actnorm=False,
                 perceptual_weight=1.0, recon_weight=1.0):
        super().__init__(disc_start, disc_num_layers, disc_in_channels, disc_weight, use_actnorm)
        if perceptual_weight > 0:
            raise WfaeError("Loss: perceptual_weight > 0 needs LPIPS (VGG16 weights fetched from the network); "
                            "not built — every shipped config sets lpips.perceptual_weight=0.0")
        self.perceptual_weight, self.recon_weight = perceptual_weight, recon_weight

    def forward(self, inputs, reconstructions, optimizer_idx, last_layer, split, global_step):
        if optimizer_idx == 1:
            d_loss, logits_real, logits_fake = self.discriminator_loss(inputs, reconstructions)
            return d_loss, {f"{split}/disc_loss": d_loss.detach(), f"{split}/logits_real": logits_real.detach().mean(),
                            f"{split}/logits_fake": logits_fake.detach().mean()}
        rec_loss = Fn.l1_loss(reconstructions, inputs, self.recon_weight)
        if global_step < self.disc_start:
            return rec_loss, {f"{split}/total_loss": rec_loss.detach(), f"{split}/rec_loss": rec_loss.detach(),
                              f"{split}/g_loss": 0.0, f"{split}/d_weight": 0.0}
        loss, g_loss, d_weight = self.generator_loss(rec_loss, reconstructions, last_layer)
        return loss, {f"{split}/total_loss": loss.detach(), f"{split}/rec_loss": rec_loss.detach(),
                      f"{split}/g_loss": g_loss.detach(), f"{split}/d_weight": d_weight}


class Model(tnn.Module):
    """reference Model (:98-214) minus Lightning: owns the autoencoder, the Loss, both optimisers and
    schedulers, and the manual-optimisation training step."""

    def __init__(self, cfg, img_size=128):
        super().__init__()
        self.cfg = cfg
        self.autoencoder = PosAwareAE_TF(img_size=img_size)
        lp = cfg.lpips
        self.loss = Loss(lp.disc_start, disc_num_layers=lp.disc_num_layers, disc_in_channels=lp.disc_in_channels,
                         disc_weight=lp.disc_weight, use_actnorm=lp.use_actnorm,
                         perceptual_weight=lp.perceptual_weight, recon_weight=lp.recon_weight)
        self.total_steps = cfg.trainer.total_train_steps
        self.accumulate_grad_batches = cfg.trainer.accumulate_grad_batches
        self.global_step = 0
        self._dp = None
        self.overlap_exchange = True    # generator all-reduce under the discriminator's forward / backward (training_step)
        self.on_after_backward = None   # optional callable(tag) run after each backward ("g" / "d"), like Lightning's hook

    def forward(self, x):
        recon, z = self.autoencoder(x)
        return recon

    def get_last_layer(self):
        return self.autoencoder.dec[-1].weight

    def configure_optimizers(self):
        """two AdamW (default betas: the reference passes only lr and weight_decay, :201,206) and two
        cosine-warmup schedulers over total_steps"""
        o, sp = self.cfg.optim, self.cfg.cosine_warmup
        warm = sp.warmup_ratio * self.total_steps
        self.g_opt = helpers.adamw_optimizer(self.autoencoder, o.lr, o.weight_decay)
        self.g_sch = helpers.cosine_warmup_scheduler(self.g_opt, sp.start_lr, sp.final_lr, sp.peak_lr, self.total_steps, warm)
        self.d_opt = helpers.adamw_optimizer(self.loss.discriminator, o.lr, o.weight_decay)
        self.d_sch = helpers.cosine_warmup_scheduler(self.d_opt, sp.start_lr, sp.final_lr, sp.peak_lr, self.total_steps, warm)
        self._dp = (parallel.DataParallelTrainer(self.autoencoder, self.g_opt),
                    parallel.DataParallelTrainer(self.loss.discriminator, self.d_opt))
        return self.g_opt, self.d_opt

    def training_step(self, batch, batch_idx=0):
        inp = batch["vil"] if isinstance(batch, dict) else batch
        clip = self.cfg.optim.gradient_clip_val
        step_now = (batch_idx + 1) % self.accumulate_grad_batches == 0
        pred = self(inp)
        logs = {}
        # ---- generator (toggle_optimizer(g_opt): the discriminator's parameters are frozen)
        with frozen(self.loss.discriminator.parameters()):
            aeloss, log_ae = self.loss(inp, pred, 0, self.get_last_layer(), "train", self.global_step)
            logs.update(log_ae)
            if self.accumulate_grad_batches != 1:
                aeloss = Fn.ScaleByFn.apply(aeloss, torch.full((), 1.0 / self.accumulate_grad_batches, device=aeloss.device))
            aeloss.backward()
        if self.on_after_backward is not None:
            self.on_after_backward("g")
        disc_on = self.global_step >= self.cfg.lpips.disc_start
        # BASELINE config 5 "overlap D/G backward with all-reduce": the generator's gradient exchange is STARTED here and
        # finished after the discriminator's forward / backward has been queued.  Legal because the discriminator's loss
        # reads the detached reconstruction and its own parameters only (reference :147-150), so the generator's clip + step
        # may wait; the arithmetic and its order per tensor are unchanged — bit-identical to the serial order
        # (tests/test_gan_gpu.py).  `overlap_exchange = False` restores the reference's literal order.
        overlap = step_now and disc_on and self.overlap_exchange
        if step_now:
            if overlap:
                self._dp[0].start_reduce()
            else:
                self._g_step(logs, clip)
        # ---- discriminator
        if disc_on:
            discloss, log_d = self.loss(inp, pred, 1, self.get_last_layer(), "train", self.global_step)
            logs.update(log_d)
            if self.accumulate_grad_batches != 1:
                discloss = Fn.ScaleByFn.apply(discloss, torch.full((), 1.0 / self.accumulate_grad_batches, device=discloss.device))
            discloss.backward()
            if self.on_after_backward is not None:
                self.on_after_backward("d")
            if overlap:
                self._g_step(logs, clip, started=True)
            if step_now:
                self._dp[1].reduce_gradients()
                logs["train/d_grad_norm"] = self.d_opt.clip_grad_norm_(clip)
                self.d_opt.step()
                self.d_sch.step()
                self.d_opt.zero_grad(set_to_none=True)
        if step_now:
            # Lightning counts one global_step per optimizer.step(); with two optimisers stepping per
            # batch the reference's disc_start comparison still sees the count of G steps before the
            # discriminator starts, which is all the comparison needs here.
            self.global_step += 1
        return pred, logs


    def _g_step(self, logs, clip, started=False):
        """generator: finish (or run) the gradient exchange, clip, step, schedule (reference :136-141)"""
        if started:
            self._dp[0].finish_reduce()
        else:
            self._dp[0].reduce_gradients()
        logs["train/g_grad_norm"] = self.g_opt.clip_grad_norm_(clip)
        self.g_opt.step()
        self.g_sch.step()
        self.g_opt.zero_grad(set_to_none=True)

    @torch.no_grad()
    def validation_step(self, batch, batch_idx=0, split="val"):
        """reference validation_step / test_step (:170-198): both halves of the loss (no gradients: d_weight is 0, as
        the reference's RuntimeError branch makes it, :74-76) and the image metrics"""
        inp = batch["vil"] if isinstance(batch, dict) else batch
        pred = self(inp)
        _, logs = self.loss(inp, pred, 0, self.get_last_layer(), split, self.global_step)
        logs = dict(logs)
        if self.global_step >= self.cfg.lpips.disc_start:
            _, log_d = self.loss(inp, pred, 1, self.get_last_layer(), split, self.global_step)
            logs.update(log_d)
        logs.update(helpers.log_metrics(pred.unsqueeze(2), inp.unsqueeze(2), split))
        return pred, logs


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--resume", type=bool, default=False)
    ap.add_argument("--config", default=os.path.join(HERE, "config.yaml"))
    ap.add_argument("--max-steps", type=int, default=-1, help="shorten the run (smoke runs): total_train_steps = this")
    ap.add_argument("--stop-after", type=int, default=-1,
                    help="stop (and checkpoint) after this many steps WITHOUT changing the schedule; `--resume True` continues")
    ap.add_argument("--matmul-precision", default="high", choices=["highest", "high", "medium"],
                    help="reference: torch.set_float32_matmul_precision('high') (train.py main).  'medium' = BASELINE "
                         "config 5's regime and MORE than torch's meaning of the word: bf16 MFMA operands AND bf16 activation "
                         "storage in HBM (the counterpart of bf16 autocast); WFAE_BF16_STORAGE=0 keeps the tensors fp32")
    args, unknown = ap.parse_known_args(argv)
    cfg = C.load(args.config, CARRIED_KEYS)
    cli = C.from_dotlist(unknown)
    helpers.check_yaml(cfg, cli)
    cfg = C.merge(cfg, cli)

    rank, world, local = parallel.init_from_env()
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    ops.set_float32_matmul_precision(args.matmul_precision)

    size, frames = (384, 49) if cfg.dataset.name == "sevir" else (128, 25)
    n_events = max(2, (cfg.dataset.batch_size * 8 * world) // (1 + (frames - cfg.dataset.seq_len) // cfg.dataset.stride) + 1)
    events = synth.blob_events(n_events, size, frames, seed=1234)
    loader = SEVIRFrameLoader(events, cfg.dataset.batch_size, cfg.dataset.seq_len, cfg.dataset.stride, "NTHW",
                              shuffle=True, device=dev, num_shard=world, rank=rank)
    accum = cfg.trainer.accumulate_grad_batches
    total = (len(loader) * cfg.trainer.max_epochs) / accum                       # reference :262
    if cfg.trainer.limit_train_batches is not None:
        total = total * cfg.trainer.limit_train_batches                          # reference :268-269
    total = max(1, int(total))
    if 0 < args.max_steps < total:
        total = args.max_steps
    cfg.trainer.total_train_steps = total
    cfg.lpips.disc_start = int(cfg.lpips.disc_start * total)                     # reference :274

    torch.manual_seed(0)
    Fn._seed_counter[0] = 0          # the counter-based dropout stream restarts with the run (restored on --resume)
    model = Model(cfg, img_size=size).to(dev).train()
    Fn.set_wgrad_overlap(True)
    model.configure_optimizers()

    ckpt_dir = os.path.join(cfg.experiment_path, "outputs", cfg.experiment_name, "checkpoints")
    last = os.path.join(ckpt_dir, "last.ckpt")
    if args.resume and os.path.exists(last):
        # Lightning restores weights, both optimisers and both schedulers (trainer.fit(ckpt_path=...), reference :296)
        ck = torch.load(last, map_location="cpu", weights_only=False)
        model.load_state_dict(ck["state_dict"], strict=True)
        model.global_step = ck.get("global_step", 0)
        if ck.get("optimizer_states"):
            model.g_opt.load_state_dict(ck["optimizer_states"][0])
            model.d_opt.load_state_dict(ck["optimizer_states"][1])
            model.g_sch.load_state_dict(ck["lr_schedulers"][0])
            model.d_sch.load_state_dict(ck["lr_schedulers"][1])
        else:
            model.g_sch.load_state_dict({"last_epoch": model.global_step})
            model.d_sch.load_state_dict({"last_epoch": max(0, model.global_step - cfg.lpips.disc_start)})
        for dp in model._dp:
            for a in dp.opt.arenas:
                dp.sync.broadcast_(a.flat_p, 0)
    stop_at = total if args.stop_after < 0 else min(total, args.stop_after)
    t0, done = time.time(), 0
    pending = []   # per-step scalars are read one step late, when they are long finished (no stall of the launch queue)

    def flush_logs():
        while pending:
            step_, logs_, lr_, n_ = pending.pop(0)
            rec = {k: float(v) for k, v in logs_.items()}
            rec.update(step=step_, lr=lr_, frames_per_s=n_ * cfg.dataset.batch_size * world / (time.time() - t0))
            print(json.dumps(rec), flush=True)

    while model.global_step < stop_at:
        first = (model.global_step * accum) % max(1, len(loader))
        for i in range(first, len(loader)):
            if model.global_step >= stop_at:
                break
            _, logs = model.training_step(loader[i], i)
            done += 1
            if rank == 0 and done % max(1, cfg.trainer.log_every_n_steps) == 0:
                flush_logs()
                pending.append((model.global_step, logs, model.g_opt.param_groups[0]["lr"], done))
    flush_logs()
    for dp in model._dp:
        dp.sync_buffers()                # collectives (one broadcast per BatchNorm buffer): every rank takes part
    if rank == 0:
        flush_bn_counters(model)
        os.makedirs(ckpt_dir, exist_ok=True)
        sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        torch.save({"state_dict": sd, "global_step": model.global_step,
                    "optimizer_states": [model.g_opt.state_dict(), model.d_opt.state_dict()],
                    "lr_schedulers": [model.g_sch.state_dict(), model.d_sch.state_dict()]}, last + ".tmp")
        os.replace(last + ".tmp", last)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()                   # nobody tears the communicator down while rank 0 still saves
        dist.destroy_process_group()
    if rank == 0:
        print("done")
    return 0


if __name__ == "__main__":
    sys.exit(main())
