"""GPU parity of the Path-B linear latent forecaster (reference
experiments/v1_experiments/pretrained_ae_linear_sevir/train.py:67,73-83) against tests/golden/g9_linear_forecast.npz
(unpinned: produced by the oracle's restatement — the reference script is not importable here)."""
import os

import numpy as np
import pytest
import torch

from tests._util import golden, relerr

pytestmark = pytest.mark.gpu


def _cfg(tin, tout, c):
    from weatherforecastingtoolkit_amd import config as C
    import weatherforecastingtoolkit_amd.experiments.v1_experiments.pretrained_ae_linear_sevir as pkg
    cfg = C.load(os.path.join(os.path.dirname(pkg.__file__), "config.yaml"))
    cfg.dataset.input_frames, cfg.dataset.pred_frames = tin, tout
    cfg.autoencoder.latent_channels = c
    cfg.trainer.total_train_steps = 20
    cfg.cosine_warmup.warmup_ratio = 0.1
    return cfg


@pytest.mark.parametrize("case", [0, 1, 2])
def test_linear_forecaster_golden(dev, case):
    from weatherforecastingtoolkit_amd import synth
    from weatherforecastingtoolkit_amd.experiments.v1_experiments.pretrained_ae_linear_sevir.train import Model
    g = golden("g9_linear_forecast")
    b, t, tin, c, h, w = [int(x) for x in g[f"{case}/cfg"]]
    model = Model(_cfg(tin, t - tin, c)).to(dev)
    with torch.no_grad():
        model.predictor.weight.copy_(torch.from_numpy(synth.uniform(9, f"lf{case}/w", tuple(model.predictor.weight.shape), -0.1, 0.1)))
        model.predictor.bias.copy_(torch.from_numpy(synth.uniform(9, f"lf{case}/b", tuple(model.predictor.bias.shape), -0.1, 0.1)))
    model.configure_optimizers()
    v = torch.from_numpy(synth.uniform(9, f"lf{case}/v", (b, t, c, h, w), -1, 1)).to(dev)
    assert relerr(model.predict_latents(v), g[f"{case}/pred_abs"]) < 2e-5
    # gradients before the step
    loss, _ = model.latent_loss(v)
    loss.backward()
    assert abs(loss.item() - float(g[f"{case}/loss"])) < 1e-5 * float(g[f"{case}/loss"])
    assert relerr(model.predictor.weight.grad, g[f"{case}/gw"]) < 2e-5
    assert relerr(model.predictor.bias.grad, g[f"{case}/gb"]) < 2e-5
    model.opt.zero_grad(set_to_none=True)
    # the whole step: clip-by-norm 1.0, AdamW, cosine-warmup
    loss, gn = model.training_step(v)
    assert abs(float(gn) - float(g[f"{case}/grad_norm"])) < 1e-5 * float(g[f"{case}/grad_norm"])
    assert relerr(model.predictor.weight, g[f"{case}/w_after"]) < 1e-6
    assert relerr(model.predictor.bias, g[f"{case}/b_after"]) < 1e-6


def test_forecaster_script_runs(dev, tmp_path):
    from weatherforecastingtoolkit_amd.experiments.v1_experiments.pretrained_ae_linear_sevir import train
    rc = train.main(["--max-steps", "3", f"experiment_path={tmp_path}", "dataset.batch_size=1"])
    assert rc == 0
    ck = torch.load(tmp_path / "outputs" / "pretrained_ae_linear_sevir" / "checkpoints" / "last.ckpt", map_location="cpu")
    assert ck["global_step"] == 3 and tuple(ck["state_dict"]["predictor.weight"].shape) == (12 * 64, 13 * 64)


def test_forecaster_on_vit_tokens(dev, tmp_path):
    """BASELINE config 4: structured latent [64, 512] (AE_ViT_2048 tokens) + linear predictor over 13 -> 12 frames"""
    from weatherforecastingtoolkit_amd.experiments.v1_experiments.pretrained_ae_linear_sevir import train
    rc = train.main(["--max-steps", "2", f"experiment_path={tmp_path}", "dataset.batch_size=1",
                     "autoencoder.kind=ae_vit.tokens", "autoencoder.latent_channels=512"])
    assert rc == 0
    ck = torch.load(tmp_path / "outputs" / "pretrained_ae_linear_sevir" / "checkpoints" / "last.ckpt", map_location="cpu")
    assert tuple(ck["state_dict"]["predictor.weight"].shape) == (12 * 512, 13 * 512)


def test_forecaster_12_in_12_out_on_64x512_latent(dev):
    """BASELINE configs[3] as stated: 12-in / 12-out frame sequences on the structured [64, 512] latent (64 tokens of
    512 channels = C 512 on an 8 x 8 latent grid).  The reference script hard-codes 13 / 12
    (v1_experiments/pretrained_ae_linear_sevir/train.py:67,73-83); the oracle's restatement follows input_frames, so the
    comparison runs against it directly on seeded latents ("parity unpinned": no reference fixture exists for it)."""
    from oracle import ae_oracle as orc
    from weatherforecastingtoolkit_amd import synth
    from weatherforecastingtoolkit_amd.experiments.v1_experiments.pretrained_ae_linear_sevir.train import Model
    tin = tout = 12
    b, c, h, w = 2, 512, 8, 8
    model = Model(_cfg(tin, tout, c)).to(dev)
    assert tuple(model.predictor.weight.shape) == (tout * c, tin * c)
    wt = torch.from_numpy(synth.uniform(11, "lf12/w", (tout * c, tin * c), -0.02, 0.02))
    bs = torch.from_numpy(synth.uniform(11, "lf12/b", (tout * c,), -0.1, 0.1))
    with torch.no_grad():
        model.predictor.weight.copy_(wt)
        model.predictor.bias.copy_(bs)
    model.configure_optimizers()
    v = torch.from_numpy(synth.uniform(11, "lf12/v", (b, tin + tout, c, h, w), -1, 1))
    wo, bo = wt.clone().requires_grad_(True), bs.clone().requires_grad_(True)
    oloss, opred = orc.linear_forecast_loss(v, wo, bo, tin)
    oloss.backward()
    vd = v.to(dev)
    assert relerr(model.predict_latents(vd), opred.detach()) < 2e-5
    loss, _ = model.latent_loss(vd)
    loss.backward()
    assert abs(loss.item() - float(oloss)) < 1e-5 * float(oloss)
    assert relerr(model.predictor.weight.grad, wo.grad) < 2e-5
    assert relerr(model.predictor.bias.grad, bo.grad) < 2e-5
