"""experiments/ae_v2 on MI355X: same CLI, YAML surface, loss and step order as the
reference's experiments/ae_v2/train.py, without Lightning / W&B.

    python -m weatherforecastingtoolkit_amd.experiments.ae_v2.train [--resume True] key=value ...

Step order (reference :209-223 + Lightning, :254-261):
  fwd -> Loss.forward (L1 [+ perceptual_weight*(1-SSIM)]) -> log -> bwd ->
  (grad all-reduce when WORLD_SIZE>1) -> AdamW -> cosine-warmup LR step.
The GAN branch of the reference Loss (:76-102) never runs in the shipped config (disc_start=1.0 =>
disc_start = total_steps, :318); it is built (experiments/_gan.py) and used when lpips.disc_start < 1.
Data: synthetic SEVIR-shaped events (synth.blob_events) pushed through the
reference's loader contract; prints `done` at the end like the reference (:347)
so its retry shell keeps working.
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
from ...nn import flush_bn_counters
from ...pipeline import helpers
from ...pipeline.datasets.sevire.sevir import SEVIRFrameLoader
from ...pipeline.models import ae_64x8x8_lin, ae_64x8x8_tf
from .._gan import GanLoss, frozen

HERE = os.path.dirname(os.path.abspath(__file__))

# Keys of the reference's experiments/ae_v2/config.yaml that are not in this build's YAML: defaults live here so that
# existing `key=value` override command lines keep parsing (helpers.check_yaml).  Read: trainer.limit_val_batches /
# limit_test_batches / save_every_n_steps, logging.log_train_all_metrics_n (metric cadence, :216-218); the rest
# (W&B / Lightning device lists / unused schedules) is accepted and ignored.
CARRIED_KEYS = {
    "project_name": "ae_test_v2",
    "lpips": {"disc_beta1": 0.5, "disc_beta2": 0.9, "disc_start_lr": 5e-7, "disc_peak_lr": 5e-6, "disc_final_lr": 5e-8,
              "disc_warmup_ratio": 0.1},
    "dataset": {"num_workers": 4, "input_frames": 0, "pred_frames": 1, "image_width": 128, "image_height": 128, "channels": 1},
    "optim": {"gradient_clip_val": 1.0},
    "one_cycle": {"peak_lr": 1e-3, "start_lr": 4e-5, "final_lr": 4e-7, "rampup_ratio": 0.3},
    "lr_range_test": {"max_lr": 1, "num_iter": 100},
    "trainer": {"devices": [0], "total_train_steps": -1, "total_val_steps": -1, "total_test_steps": -1,
                "save_every_n_steps": 0.1, "save_on_train_epoch_end": False, "limit_train_batches": 0.01,
                "limit_val_batches": 0.01, "limit_test_batches": 0.01},
    "logging": {"wandb_watch_log_freq": 0, "log_train_all_metrics_n": 0.01, "log_train_plots_n": 0.01, "log_val_plots_n": 0.01},
}



class Loss(GanLoss):
    """reference Loss (experiments/ae_v2/train.py:27-102): L1 [+ perceptual_weight * (1 - SSIM)], and past `disc_start`
    the generator term d_weight * -mean(D(x_hat)) (optimizer_idx 0) / the hinge discriminator loss (optimizer_idx 1).
    Owns the PatchGAN discriminator and `logvar` like the reference (state_dict keys `discriminator.main.*`,
    `logvar`); LPIPS is commented out in the reference (:60-61) and absent here."""

    def __init__(self, disc_start, disc_num_layers=3, disc_in_channels=1, disc_weight=1.0, use_actnorm=False,
                 perceptual_weight=1.0, kl_weight=1.0, logvar_init=0.0, recon_weight=1.0):
        super().__init__(disc_start, disc_num_layers, disc_in_channels, disc_weight, use_actnorm)
        self.perceptual_weight, self.recon_weight, self.kl_weight = perceptual_weight, recon_weight, kl_weight
        self.logvar = tnn.Parameter(torch.ones(size=()) * logvar_init)      # unused by the live branch (:42)

    def forward(self, inputs, reconstructions, posteriors=None, optimizer_idx=0, last_layer=None,
                split="train", global_step=0):
        if optimizer_idx == 1:
            d_loss, logits_real, logits_fake = self.discriminator_loss(inputs, reconstructions)
            return d_loss, {f"{split}/disc_loss": d_loss.detach(), f"{split}/logits_real": logits_real.detach().mean(),
                            f"{split}/logits_fake": logits_fake.detach().mean()}
        rec_loss = Fn.l1_loss(reconstructions, inputs, self.recon_weight)
        if self.perceptual_weight > 0:
            # 1 - ssim on channel-tripled inputs (:57-63); tripling does not change the value
            s = Fn.ssim(inputs, reconstructions)
            rec_loss = rec_loss + self.perceptual_weight * (1.0 - s)  # 0-dim scalar arithmetic
        nll_loss = rec_loss
        if global_step < self.disc_start:
            return nll_loss, {f"{split}/total_loss": nll_loss.detach(), f"{split}/rec_loss": rec_loss.detach(),
                              f"{split}/nll_loss": nll_loss.detach(), f"{split}/g_loss": 0.0, f"{split}/d_weight": 0.0}
        loss, g_loss, d_weight = self.generator_loss(nll_loss, reconstructions, last_layer)
        return loss, {f"{split}/total_loss": loss.detach(), f"{split}/nll_loss": nll_loss.detach(),
                      f"{split}/rec_loss": rec_loss.detach(), f"{split}/g_loss": g_loss.detach(), f"{split}/d_weight": d_weight}


class Model(tnn.Module):
    """reference Model (experiments/ae_v2/train.py:181-261) minus Lightning: owns the autoencoder and the Loss, the
    three step functions with the reference's logging keys and metric cadence, and the optimiser / scheduler pair
    (`configure_optimizers`, :254-261: only `self.autoencoder.parameters()` are optimised)."""

    def __init__(self, cfg, img_size=128, variant="tf"):
        super().__init__()
        self.cfg = cfg
        model_mod = ae_64x8x8_tf if variant == "tf" else ae_64x8x8_lin
        self.autoencoder = model_mod.PosAwareAE_TF(img_size=img_size)
        lp = cfg.lpips
        self.loss = Loss(lp.disc_start, disc_num_layers=lp.disc_num_layers, disc_in_channels=lp.disc_in_channels,
                         disc_weight=lp.disc_weight, use_actnorm=lp.use_actnorm,
                         perceptual_weight=lp.perceptual_weight, kl_weight=lp.kl_weight,
                         logvar_init=lp.logvar_init, recon_weight=lp.recon_weight)
        self.total_steps = cfg.trainer.total_train_steps
        self.global_step = 0
        self.current_epoch = 0

    def forward(self, x):
        return self.autoencoder(x)

    def get_last_layer(self):
        return self.autoencoder.dec[-1].weight

    def configure_optimizers(self):
        o, sp = self.cfg.optim, self.cfg.cosine_warmup
        self.opt = helpers.adamw_optimizer(self.autoencoder, o.lr, o.weight_decay, o.beta1, o.beta2)
        self.sched = helpers.cosine_warmup_scheduler(self.opt, sp.start_lr, sp.final_lr, sp.peak_lr, self.total_steps,
                                                     sp.warmup_ratio * self.total_steps)
        return self.opt, self.sched

    def training_step(self, batch, batch_idx):
        """:209-223 — returns (loss, scalar logs).  Image metrics every log_train_all_metrics_n * total_train_steps
        batches (:216-218), under the reference's `train_*` names."""
        inp = batch["vil"]
        pred, z = self(inp)
        # the reference trains only the autoencoder here: past disc_start the discriminator scores the
        # reconstruction but is never updated — its parameters stay frozen
        with frozen(self.loss.discriminator.parameters()):
            aeloss, logs = self.loss(inp, pred, z, 0, self.get_last_layer(), "train", self.global_step)
        logs = dict(logs)
        interval = max(1, int(self.cfg.logging.log_train_all_metrics_n * self.cfg.trainer.total_train_steps))
        if batch_idx % interval == 0:
            logs.update(helpers.log_metrics(pred.unsqueeze(2), inp.unsqueeze(2), "train"))
        return aeloss, logs

    @torch.no_grad()
    def validation_step(self, batch, batch_idx):
        """:225-236: generator half of the loss + the image metrics on every batch"""
        inp = batch["vil"]
        pred, z = self(inp)
        aeloss, logs = self.loss(inp, pred, z, 0, self.get_last_layer(), "val", self.global_step)
        logs = dict(logs)
        logs.update(helpers.log_metrics(pred.unsqueeze(2), inp.unsqueeze(2), "val"))
        return aeloss, logs

    @torch.no_grad()
    def test_step(self, batch, batch_idx):
        """:238-252: both halves of the loss + the image metrics"""
        inp = batch["vil"]
        pred, z = self(inp)
        _, logs = self.loss(inp, pred, z, 0, self.get_last_layer(), "test", self.global_step)
        logs = dict(logs)
        _, log_d = self.loss(inp, pred, z, 1, self.get_last_layer(), "test", self.global_step)
        logs.update(log_d)
        logs.update(helpers.log_metrics(pred.unsqueeze(2), inp.unsqueeze(2), "test"))
        return logs


def _mean_logs(rows, world):
    """epoch mean of per-step scalar logs (Lightning on_epoch=True), averaged over ranks (sync_dist=True)"""
    if world > 1:
        # every rank enters the same collectives even when its shard of the loader was empty (ranks would otherwise wait
        # for ever): the key set is agreed first, then sums and counts travel together
        import torch.distributed as dist
        gathered = [None] * world
        dist.all_gather_object(gathered, sorted({k for r in rows for k in r}))
        keys = sorted({k for ks in gathered for k in ks})
        if not keys:
            return {}
        t = torch.tensor([[sum(float(r[k]) for r in rows if k in r) for k in keys],
                          [float(sum(1 for r in rows if k in r)) for k in keys]], dtype=torch.float64).cuda()
        dist.all_reduce(t)
        t = t.cpu()
        # mean over the rows of all ranks (= the mean of the rank means of sync_dist=True when the shards are equal; a rank
        # without rows contributes nothing)
        return {k: float(t[0, i] / max(1.0, float(t[1, i]))) for i, k in enumerate(keys)}
    if not rows:
        return {}
    keys = sorted({k for r in rows for k in r})
    return {k: sum(float(r[k]) for r in rows if k in r) / max(1, sum(1 for r in rows if k in r)) for k in keys}


def evaluate(model, loader, split, limit, world):
    """one pass over `limit` (fraction, Lightning limit_*_batches) of the loader in eval mode"""
    net_training = model.training
    model.eval()
    n = len(loader)
    n = n if limit is None else max(1, int(n * limit)) if n else 0
    rows = []
    for i in range(n):
        batch = loader[i]
        rows.append(model.test_step(batch, i) if split == "test" else model.validation_step(batch, i)[1])
    model.train(net_training)
    out = _mean_logs(rows, world)
    out["batches"] = n
    return out


def save_checkpoint(path, model, epoch):
    """Lightning .ckpt layout (keys `state_dict` with the `autoencoder.` / `loss.` prefixes of the reference's
    LightningModule attributes, `optimizer_states`, `lr_schedulers`, `global_step`, `epoch`)"""
    flush_bn_counters(model)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    tmp = path + ".tmp"
    torch.save({"state_dict": sd, "global_step": model.global_step, "epoch": epoch,
                "optimizer_states": [model.opt.state_dict()], "lr_schedulers": [model.sched.state_dict()],
                "rng": {"torch_seed": torch.initial_seed(), "dropout_counter": Fn._seed_counter[0],
                        "torch_state": torch.get_rng_state(), "cuda_state": torch.cuda.get_rng_state()}}, tmp)
    os.replace(tmp, path)


def load_checkpoint(path, model):
    ck = torch.load(path, map_location="cpu", weights_only=False)
    sd = ck["state_dict"]
    if not any(k.startswith("loss.") for k in sd):      # AE-only checkpoints (round-1 files, exported weights)
        sd = dict(sd)
        sd.update({k: v for k, v in model.state_dict().items() if k.startswith("loss.")})
    model.load_state_dict(sd, strict=True)
    model.global_step = int(ck.get("global_step", 0))
    if ck.get("optimizer_states"):
        model.opt.load_state_dict(ck["optimizer_states"][0])
    else:   # weights-only checkpoint: moments restart from zero, the bias correction restarts with them
        print("checkpoint holds no optimiser state: AdamW moments restart from zero", file=sys.stderr)
    model.sched.load_state_dict(ck["lr_schedulers"][0] if ck.get("lr_schedulers") else {"last_epoch": model.global_step})
    rng = ck.get("rng")
    if rng:
        torch.manual_seed(rng["torch_seed"])
        if rng.get("torch_state") is not None:      # the generators continue where the interrupted run stopped
            torch.set_rng_state(rng["torch_state"])
            torch.cuda.set_rng_state(rng["cuda_state"])
        Fn._seed_counter[0] = rng["dropout_counter"]
    return int(ck.get("epoch", 0))


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--resume", type=bool, default=False)
    ap.add_argument("--config", default=os.path.join(HERE, "config.yaml"))
    ap.add_argument("--max-steps", type=int, default=-1, help="shorten the run (smoke runs): total_train_steps = this")
    ap.add_argument("--stop-after", type=int, default=-1,
                    help="stop (and checkpoint) after this many optimiser steps WITHOUT changing the schedule — an "
                         "interrupted run that `--resume True` continues")
    ap.add_argument("--matmul-precision", default="high", choices=["highest", "high", "medium"],
                    help="reference: torch.set_float32_matmul_precision('high') (train.py main).  'medium' = BASELINE "
                         "config 5's regime and MORE than torch's meaning of the word: bf16 MFMA operands AND bf16 activation "
                         "storage in HBM (the counterpart of bf16 autocast); WFAE_BF16_STORAGE=0 keeps the tensors fp32")
    ap.add_argument("--data-dir", default=None, help="SEVIR root (CATALOG.csv + data/); default: synthetic events")
    ap.add_argument("--data-format", choices=("npy", "h5"), default="npy")
    ap.add_argument("--model", choices=("tf", "lin"), default="tf",
                    help="tf = ae_64x8x8_tf (what the reference ae_v2/train.py:18 imports), lin = ae_64x8x8_lin (ae_v2_2)")
    ap.add_argument("--test", action="store_true", help="run the test loop after training (trainer.test)")
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
    B = cfg.dataset.batch_size
    nspe = 1 + (frames - cfg.dataset.seq_len) // cfg.dataset.stride
    if args.data_dir:
        # real data: <data_dir>/CATALOG.csv + the event files (.npy, or .h5 when h5py is available), split by date
        # like the reference (train < 2019-01-01 <= val < 2019-06-01 <= test, sevire/sevir.py:1227-1243)
        import datetime
        from ...pipeline.datasets.sevire.catalog import CatalogEventStore, H5EventSource, NpyEventSource, SEVIRCatalog
        cat_path = os.path.join(args.data_dir, "CATALOG.csv")
        source = (H5EventSource if args.data_format == "h5" else NpyEventSource)(os.path.join(args.data_dir, "data"))
        d_val, d_test = datetime.datetime(2019, 1, 1), datetime.datetime(2019, 6, 1)
        def store(**kw):
            cat = SEVIRCatalog(cat_path, **kw)
            return CatalogEventStore(cat, source) if len(cat) else None       # an empty split: that loop is skipped

        ev_train = store(end_date=d_val, shuffle=True, shuffle_seed=1)
        ev_val = store(start_date=d_val, end_date=d_test)
        ev_test = store(start_date=d_test) if args.test else None     # opened only when the test loop will run
        if ev_train is None:
            raise ValueError(f"{cat_path}: no training events before {d_val:%Y-%m-%d}")
        size = ev_train.event_shape[0]
    else:
        n_events = max(2, (B * 8 * world) // nspe + 1)
        ev_train = synth.blob_events(n_events, size, frames, seed=1234)
        ev_val = synth.blob_events(max(1, (B * 2 * world) // nspe + 1), size, frames, seed=4321)
        ev_test = synth.blob_events(max(1, (B * 2 * world) // nspe + 1), size, frames, seed=9876)

    def mk(ev, shuffle):
        if ev is None:
            return ()
        return SEVIRFrameLoader(ev, B, cfg.dataset.seq_len, cfg.dataset.stride, "NTHW", shuffle=shuffle, device=dev,
                                num_shard=world, rank=rank)
    loader, val_loader, test_loader = mk(ev_train, True), mk(ev_val, False), mk(ev_test, False)
    accum = cfg.trainer.accumulate_grad_batches
    total_steps = max(1, int(len(loader) * cfg.trainer.max_epochs / accum))  # reference :306
    if 0 < args.max_steps < total_steps:
        total_steps = args.max_steps
    cfg.trainer.total_train_steps = total_steps
    cfg.trainer.total_val_steps = max(1, int(len(val_loader) * cfg.trainer.max_epochs / accum))
    cfg.trainer.total_test_steps = max(1, int(len(test_loader) * cfg.trainer.max_epochs / accum))
    cfg.lpips.disc_start = int(cfg.lpips.disc_start * total_steps)            # reference :318

    torch.manual_seed(0)
    Fn._seed_counter[0] = 0          # the counter-based dropout stream restarts with the run (restored on --resume)
    model = Model(cfg, img_size=size, variant=args.model).to(dev).train()
    net, loss_fn = model.autoencoder, model.loss
    Fn.set_wgrad_overlap(True)
    opt, sched = model.configure_optimizers()
    dp = parallel.DataParallelTrainer(net, opt)

    ckpt_dir = os.path.join(cfg.experiment_path, "outputs", cfg.experiment_name, "checkpoints")
    last = os.path.join(ckpt_dir, "last.ckpt")
    epoch = 0
    if args.resume and os.path.exists(last):
        epoch = load_checkpoint(last, model)
        for a in opt.arenas:
            dp.sync.broadcast_(a.flat_p, 0)
    save_every = max(1, int(total_steps * cfg.trainer.save_every_n_steps))   # helpers.py:241
    stop_at = total_steps if args.stop_after < 0 else min(total_steps, args.stop_after)
    nb = len(loader)
    t0, t_steps = time.time(), 0
    # Per-step logging (trainer.log_every_n_steps = 1 in the reference's config) without a per-step stall: the scalars of
    # step s stay on the device until step s + 1 has been queued, then they are read — by then they are long finished and
    # the launch queue never runs dry (reading them right away cost ~40 ms of a 210 ms step at B = 32, 384x384).
    pending = []

    def flush_logs():
        while pending:
            step_, logs_, lr_, n_ = pending.pop(0)
            rec = {k: float(v) for k, v in logs_.items()}
            rec.update(step=step_, lr=lr_, frames_per_s=n_ * B * world / (time.time() - t0))
            print(json.dumps(rec), flush=True)

    while model.global_step < stop_at:
        model.current_epoch = epoch = model.global_step // max(1, nb)
        first = model.global_step % max(1, nb)                                # mid-epoch resume: skip the batches done
        for batch_idx, batch in enumerate(loader.prefetch(2, start=first), start=first):
            if model.global_step >= stop_at:
                break
            opt.zero_grad(set_to_none=True)
            loss, logs = model.training_step(batch, batch_idx)
            loss.backward()
            dp.reduce_gradients()
            opt.step()
            sched.step()
            model.global_step += 1
            t_steps += 1
            step = model.global_step
            if rank == 0 and step % max(1, cfg.trainer.log_every_n_steps) == 0:
                flush_logs()
                pending.append((step, logs, opt.param_groups[0]["lr"], t_steps))
            if step % save_every == 0 and step < stop_at:
                dp.sync_buffers(model)                                             # collective: every rank
                if rank == 0:
                    save_checkpoint(last, model, epoch)
        flush_logs()
        if model.global_step % max(1, nb) == 0 or model.global_step >= total_steps:
            # Lightning runs the validation loop at the end of every training epoch
            dp.sync_buffers(model)
            vlog = evaluate(model, val_loader, "val", cfg.trainer.limit_val_batches, world)
            if rank == 0 and vlog.get("batches"):
                vlog.update(step=model.global_step, epoch=epoch)
                print(json.dumps(vlog), flush=True)
    flush_logs()
    dp.sync_buffers(model)                    # a collective (one broadcast per BatchNorm buffer): every rank takes part
    if args.test:
        tlog = evaluate(model, test_loader, "test", cfg.trainer.limit_test_batches, world)
        if rank == 0:
            print(json.dumps(tlog), flush=True)
    if rank == 0:
        save_checkpoint(last, model, epoch)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()                   # nobody leaves (and tears the communicator down) while rank 0 still saves
        dist.destroy_process_group()
    if rank == 0:
        print("done")
    return 0


if __name__ == "__main__":
    sys.exit(main())
