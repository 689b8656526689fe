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

# Keys of the reference's experiments/ae_v2/config.yaml that this build accepts (so that existing `key=value`
# override command lines keep parsing, helpers.check_yaml) but never reads: W&B / Lightning / unused schedules.
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


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--resume", type=bool, default=False)
    ap.add_argument("--config", default=os.path.join(HERE, "config.yaml"))
    ap.add_argument("--max-steps", type=int, default=-1, help="stop early (smoke runs)")
    ap.add_argument("--matmul-precision", default="high", choices=["highest", "high", "medium"],
                    help="reference: torch.set_float32_matmul_precision('high') (train.py main); 'medium' = bf16 "
                         "MFMA operands (BASELINE config 5)")
    ap.add_argument("--data-dir", default=None, help="SEVIR root (CATALOG.csv + data/); default: synthetic events")
    ap.add_argument("--data-format", choices=("npy", "h5"), default="npy")
    ap.add_argument("--model", choices=("tf", "lin"), default="tf",
                    help="tf = ae_64x8x8_tf (what the reference ae_v2/train.py:18 imports), lin = ae_64x8x8_lin (ae_v2_2)")
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
    if args.data_dir:
        # real data: <data_dir>/CATALOG.csv + the event files (.npy, or .h5 when h5py is available), the
        # reference's train split (events before 2019-06-01, pipeline/datasets/sevire/sevir.py:1089-1099)
        import datetime
        from ...pipeline.datasets.sevire.catalog import CatalogEventStore, H5EventSource, NpyEventSource, SEVIRCatalog
        catalog = SEVIRCatalog(os.path.join(args.data_dir, "CATALOG.csv"), end_date=datetime.datetime(2019, 6, 1),
                               shuffle=True, shuffle_seed=1)
        source = (H5EventSource if args.data_format == "h5" else NpyEventSource)(os.path.join(args.data_dir, "data"))
        events = CatalogEventStore(catalog, source)
        size = events.event_shape[0]
    else:
        n_events = max(2, (cfg.dataset.batch_size * 8 * world) // (1 + (frames - cfg.dataset.seq_len) // cfg.dataset.stride) + 1)
        events = synth.blob_events(n_events, size, frames, seed=1234)
    loader = SEVIRFrameLoader(events, cfg.dataset.batch_size, cfg.dataset.seq_len, cfg.dataset.stride, "NTHW",
                              shuffle=True, device=dev, num_shard=world, rank=rank)
    accum = cfg.trainer.accumulate_grad_batches
    total_steps = max(1, int(len(loader) * cfg.trainer.max_epochs / accum))  # reference :306
    if 0 < args.max_steps < total_steps:
        total_steps = args.max_steps
    disc_start = int(cfg.lpips.disc_start * total_steps)                      # reference :318

    torch.manual_seed(0)
    model_mod = ae_64x8x8_tf if args.model == "tf" else ae_64x8x8_lin
    net = model_mod.PosAwareAE_TF(img_size=size).to(dev).train()
    Fn.set_wgrad_overlap(True)
    loss_fn = Loss(disc_start, disc_num_layers=cfg.lpips.disc_num_layers, disc_in_channels=cfg.lpips.disc_in_channels,
                   disc_weight=cfg.lpips.disc_weight, use_actnorm=cfg.lpips.use_actnorm,
                   perceptual_weight=cfg.lpips.perceptual_weight, kl_weight=cfg.lpips.kl_weight,
                   logvar_init=cfg.lpips.logvar_init, recon_weight=cfg.lpips.recon_weight).to(dev)
    opt = helpers.adamw_optimizer(net, cfg.optim.lr, cfg.optim.weight_decay, cfg.optim.beta1, cfg.optim.beta2)
    sched = helpers.cosine_warmup_scheduler(opt, cfg.cosine_warmup.start_lr, cfg.cosine_warmup.final_lr,
                                            cfg.cosine_warmup.peak_lr, total_steps,
                                            cfg.cosine_warmup.warmup_ratio * total_steps)
    dp = parallel.DataParallelTrainer(net, opt)

    ckpt_dir = os.path.join(cfg.experiment_path, "outputs", cfg.experiment_name, "checkpoints")
    step = 0
    last = os.path.join(ckpt_dir, "last.ckpt")
    if args.resume and os.path.exists(last):
        ck = torch.load(last, map_location="cpu")
        sd = {k[len("autoencoder."):]: v for k, v in ck["state_dict"].items() if k.startswith("autoencoder.")}
        net.load_state_dict(sd, strict=True)
        step = ck.get("global_step", 0)
        sched.load_state_dict({"last_epoch": step})
    t0 = time.time()
    while step < total_steps:
        for batch in loader.prefetch(2):
            if step >= total_steps:
                break
            inp = batch["vil"]
            opt.zero_grad(set_to_none=True)
            pred, z = net(inp)
            # the reference trains only the autoencoder here (one optimiser, :254-261): past disc_start the
            # discriminator scores the reconstruction but is never updated — its parameters stay frozen
            with frozen(loss_fn.discriminator.parameters()):
                loss, logs = loss_fn(inp, pred, None, 0, net.dec[-1].weight, "train", step)
                loss.backward()
            dp.reduce_gradients()
            opt.step()
            sched.step()
            step += 1
            if rank == 0 and step % max(1, cfg.trainer.log_every_n_steps) == 0:
                rec = {k: float(v) for k, v in logs.items()}
                rec.update(step=step, lr=opt.param_groups[0]["lr"], frames_per_s=step * cfg.dataset.batch_size * world / (time.time() - t0))
                print(json.dumps(rec), flush=True)
    if rank == 0:
        dp.sync_buffers()
        flush_bn_counters(net)
        os.makedirs(ckpt_dir, exist_ok=True)
        sd = {"autoencoder." + k: v.detach().cpu() for k, v in net.state_dict().items()}
        torch.save({"state_dict": sd, "global_step": step}, last)
        print("done")
    return 0


if __name__ == "__main__":
    sys.exit(main())
