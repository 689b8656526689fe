"""Path-B linear latent forecaster on MI355X: the predictor step of the reference's
experiments/v1_experiments/pretrained_ae_linear_sevir/train.py without Lightning / W&B.

    python -m weatherforecastingtoolkit_amd.experiments.v1_experiments.pretrained_ae_linear_sevir.train key=value ...

Reference step (:73-83): latents v (B,T,C,h,w) of a frozen autoencoder; inp = v[:, :Tin] - v[:, Tin-1],
tgt = v[:, Tin:] - v[:, Tin-1]; pred = Linear(Tin*C -> Tout*C) applied per latent pixel on the
(b,h,w,Tin*C) layout; loss = mse(pred, tgt); AdamW + cosine-warmup on the predictor only, clip 1.0 (:189).
Here: one differencing/layout kernel, one MFMA GEMM (+ its weight-gradient GEMM), one MSE kernel.
The frozen AutoencoderKL of the reference (pretrained checkpoint, not available) is replaced by a pluggable
latent provider; by default the frozen `enc` stack of the ae_v2 conv autoencoder.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch
import torch.nn as tnn

from .... import config as C
from .... import functional as Fn
from .... import nn as wnn
from .... import ops, parallel, synth
from ....pipeline import helpers
from ....pipeline.datasets.sevire.sevir import SEVIRFrameLoader
from ....pipeline.models.ae_64x8x8_lin import PosAwareAE_TF

HERE = os.path.dirname(os.path.abspath(__file__))


class Autoencoder(tnn.Module):
    """frozen latent provider with the reference wrapper's interface (:21-56): encode (B,T,1,H,W) -> (B,T,C,h,w).
    kind "ae_64x8x8_lin.enc": the conv encoder stack (64 channels at 1/16 resolution);
    kind "ae_vit.tokens": the structured token latent [64, 512] of AE_ViT_2048 (BASELINE config 4), i.e. the
    encoder tokens laid out as 512 channels on the 8x8 patch grid."""

    def __init__(self, img_size=128, kind="ae_64x8x8_lin.enc"):
        super().__init__()
        self.kind = kind
        if kind == "ae_vit.tokens":
            from ....pipeline.models.ae_vit import AE_ViT_2048
            self.autoencoder = AE_ViT_2048().eval()
        elif kind == "ae_64x8x8_lin.enc":
            self.autoencoder = PosAwareAE_TF(img_size=img_size).eval()
        else:
            raise ValueError(f"autoencoder.kind={kind!r}")
        for p in self.autoencoder.parameters():
            p.requires_grad_(False)

    @torch.no_grad()
    def encode(self, x):
        b, t, c, h, w = x.shape
        frames = x.reshape(b * t, c, h, w)
        if self.kind == "ae_vit.tokens":
            tok = self.autoencoder.encode_tokens(frames)                      # (B*T, 64, 512)
            s = self.autoencoder.seq
            z = tok.transpose(1, 2).contiguous().view(b * t, tok.shape[2], s, s)
        else:
            z = self.autoencoder.enc(frames)
        return z.view(b, t, *z.shape[1:])


class Model(tnn.Module):
    """reference Model (:58-134): `predictor`, `forward`, the training step and its optimiser"""

    def __init__(self, cfg, latent_channels=None, autoencoder=None):
        super().__init__()
        self.cfg = cfg
        self.autoencoder = autoencoder
        self.input_frames, self.pred_frames = cfg.dataset.input_frames, cfg.dataset.pred_frames
        c = latent_channels if latent_channels is not None else cfg.autoencoder.latent_channels
        self.latent_channels = c
        self.predictor = wnn.Linear(self.input_frames * c, self.pred_frames * c)
        self.total_steps = cfg.trainer.total_train_steps

    def forward(self, x):
        return self.predictor(x)

    def latent_loss(self, v):
        """v (B,T,C,h,w) latents -> (loss, pred (B*h*w, Tout*C)) — reference :75-82"""
        X, Y = ops.latent_diff_pack(v.contiguous(), self.input_frames)
        pred = self(X)
        return Fn.mse_loss(pred, Y), pred

    def predict_latents(self, v):
        """forecast latents (B,Tout,C,h,w) = pred + last input frame (reference :86-87)"""
        X, _ = ops.latent_diff_pack(v.contiguous(), self.input_frames)
        with torch.no_grad():
            return ops.latent_unpack_add(self(X).contiguous(), v.contiguous(), self.input_frames)

    def configure_optimizers(self):
        o, sp = self.cfg.optim, self.cfg.cosine_warmup
        self.opt = helpers.adamw_optimizer(self.predictor, o.lr, o.weight_decay)
        self.sch = helpers.cosine_warmup_scheduler(self.opt, sp.start_lr, sp.final_lr, sp.peak_lr, self.total_steps,
                                                   sp.warmup_ratio * self.total_steps)
        self._dp = parallel.DataParallelTrainer(self.predictor, self.opt)
        return self.opt

    def training_step(self, batch, batch_idx=0):
        """batch: frames (B,T,H,W) fp32 in [0,1] ('NTHW') or latents (B,T,C,h,w)"""
        if batch.dim() == 4:
            v = self.autoencoder.encode(batch.unsqueeze(2))
        else:
            v = batch
        loss, _ = self.latent_loss(v)
        loss.backward()
        self._dp.reduce_gradients()
        gn = self.opt.clip_grad_norm_(self.cfg.optim.gradient_clip_val)
        self.opt.step()
        self.sch.step()
        self.opt.zero_grad(set_to_none=True)
        return loss.detach(), gn


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default=os.path.join(HERE, "config.yaml"))
    ap.add_argument("--max-steps", type=int, default=-1)
    args, unknown = ap.parse_known_args(argv)
    cfg = C.load(args.config)
    cli = C.from_dotlist(unknown)
    helpers.check_yaml(cfg, cli)
    cfg = C.merge(cfg, cli)
    rank, world, local = parallel.init_from_env()
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    size, frames = (384, 49) if cfg.dataset.name == "sevir" else (128, 25)
    events = synth.blob_events(max(2, cfg.dataset.batch_size * 2 * world), size, frames, seed=1234)
    loader = SEVIRFrameLoader(events, cfg.dataset.batch_size, cfg.dataset.seq_len, cfg.dataset.stride, "NTHW",
                              shuffle=True, device=dev, num_shard=world, rank=rank)
    total = max(1, int(len(loader) * cfg.trainer.max_epochs / cfg.trainer.accumulate_grad_batches))
    if 0 < args.max_steps < total:
        total = args.max_steps
    cfg.trainer.total_train_steps = total
    torch.manual_seed(0)
    model = Model(cfg, autoencoder=Autoencoder(size, cfg.autoencoder.kind)).to(dev).train()
    model.autoencoder.eval()
    model.configure_optimizers()
    step, t0 = 0, time.time()
    while step < total:
        for batch in loader:
            if step >= total:
                break
            loss, gn = model.training_step(batch["vil"])
            step += 1
            if rank == 0 and step % max(1, cfg.trainer.log_every_n_steps) == 0:
                print(json.dumps({"step": step, "train_loss": float(loss), "grad_norm": float(gn),
                                  "lr": model.opt.param_groups[0]["lr"],
                                  "sequences_per_s": step * cfg.dataset.batch_size * world / (time.time() - t0)}), flush=True)
    if rank == 0:
        out = os.path.join(cfg.experiment_path, "outputs", cfg.experiment_name, "checkpoints")
        os.makedirs(out, exist_ok=True)
        torch.save({"state_dict": {"predictor." + k: v.detach().cpu() for k, v in model.predictor.state_dict().items()},
                    "global_step": step}, os.path.join(out, "last.ckpt"))
        print("done")
    return 0


if __name__ == "__main__":
    sys.exit(main())
