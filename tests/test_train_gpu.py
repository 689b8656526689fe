"""GPU: the experiment scripts end to end — checkpoint/resume equivalence, validation / test steps with the
reference's metric names, and both train scripts with two data-parallel ranks sharing the card over gloo."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _params(path):
    ck = torch.load(path, map_location="cpu", weights_only=False)
    return ck


@pytest.mark.parametrize("variant", ["tf", "lin"])
def test_resume_reproduces_uninterrupted_run(dev, tmp_path, variant):
    """train 3 steps, save, resume, train the 4th: parameters, BatchNorm buffers, AdamW moments and the LR are
    bit-identical to an uninterrupted 4-step run (reference: Lightning restores optimiser + scheduler state,
    experiments/ae_v2/train.py:322,346; `tf` includes the dropout stream of the latent transformer)"""
    from weatherforecastingtoolkit_amd.experiments.ae_v2 import train
    common = ["--model", variant, "--max-steps", "6", "dataset.batch_size=2"]
    a, b = tmp_path / "a", tmp_path / "b"
    assert train.main(common + ["--stop-after", "4", f"experiment_path={a}"]) == 0
    assert train.main(common + ["--stop-after", "3", f"experiment_path={b}"]) == 0
    mid = _params(b / "outputs" / "ae_2048_attn" / "checkpoints" / "last.ckpt")
    assert mid["global_step"] == 3 and mid["optimizer_states"] and mid["lr_schedulers"][0]["last_epoch"] == 3
    assert train.main(common + ["--stop-after", "4", "--resume", "True", f"experiment_path={b}"]) == 0
    ca = _params(a / "outputs" / "ae_2048_attn" / "checkpoints" / "last.ckpt")
    cb = _params(b / "outputs" / "ae_2048_attn" / "checkpoints" / "last.ckpt")
    assert ca["global_step"] == cb["global_step"] == 4
    assert ca["state_dict"].keys() == cb["state_dict"].keys()
    assert any(k.startswith("autoencoder.") for k in ca["state_dict"]) and "loss.logvar" in ca["state_dict"]
    for k in ca["state_dict"]:
        assert torch.equal(ca["state_dict"][k], cb["state_dict"][k]), k
    sa, sb = ca["optimizer_states"][0], cb["optimizer_states"][0]
    assert sa["param_groups"] == sb["param_groups"]
    assert torch.equal(sa["state"][0]["exp_avg"], sb["state"][0]["exp_avg"])
    assert torch.equal(sa["state"][0]["exp_avg_sq"], sb["state"][0]["exp_avg_sq"])
    assert float(sa["state"][0]["exp_avg_sq"].abs().sum()) > 0


def test_validation_and_test_steps(dev):
    """validation_step / test_step of the reference Model (:225-252): loss keys under the split's name and the image
    metrics `{split}_paper_SSIM` / `{split}_paper_PSNR`, equal to the kernels' values on the same tensors"""
    from weatherforecastingtoolkit_amd import config as C, ops, synth
    from weatherforecastingtoolkit_amd.experiments.ae_v2 import train
    cfg = C.load(os.path.join(os.path.dirname(train.__file__), "config.yaml"), train.CARRIED_KEYS)
    cfg.trainer.total_train_steps = 10
    cfg.lpips.disc_start = 10
    torch.manual_seed(0)
    model = train.Model(cfg, img_size=128, variant="lin").to(dev).eval()
    x = torch.from_numpy(synth.uniform_frames(3, 128, seed=3)).to(dev)
    loss, logs = model.validation_step({"vil": x}, 0)
    with torch.no_grad():
        pred, _ = model(x)
    s = ops.ssim_fwd(x, pred, clamp01=True).item()
    p = ops.psnr(pred, x, clamp01=True).item()
    assert logs["val_paper_SSIM"] == s and logs["val_paper_PSNR"] == p
    assert abs(float(logs["val/rec_loss"]) - (pred - x).abs().mean().item()) < 1e-6
    assert float(loss) == float(logs["val/total_loss"])
    tlogs = model.test_step({"vil": x}, 0)
    for k in ("test/rec_loss", "test/disc_loss", "test/logits_real", "test/logits_fake", "test_paper_SSIM", "test_paper_PSNR"):
        assert k in tlogs, k
    # training_step: metrics only at the reference cadence (every int(0.01 * total) = 1 -> each batch here; every 5th below)
    model.train()
    cfg.logging.log_train_all_metrics_n = 0.5
    _, l0 = model.training_step({"vil": x}, 0)
    _, l1 = model.training_step({"vil": x}, 1)
    assert "train_paper_SSIM" in l0 and "train_paper_SSIM" not in l1 and "train/rec_loss" in l1


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("script", ["ae_v2", "ae_v2_2"])
def test_train_script_two_ranks_finishes(dev, tmp_path, script):
    """both train scripts with WORLD_SIZE=2 (two ranks on the one card, gloo): every rank reaches the end (the buffer
    broadcast before the checkpoint is a collective), rank 0 writes last.ckpt and prints `done`"""
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE="2",
                   LOCAL_RANK="0", WFAE_DIST_BACKEND="gloo", PYTHONPATH=ROOT)
        cmd = [sys.executable, "-m", f"weatherforecastingtoolkit_amd.experiments.{script}.train", "--max-steps", "2",
               f"experiment_path={tmp_path}", "dataset.batch_size=2"]
        if script == "ae_v2":
            cmd += ["--model", "lin"]
        else:
            cmd += ["lpips.disc_start=0.0", "lpips.disc_weight=0.5"]
        procs.append(subprocess.Popen(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=420)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail(f"{script}: a rank hung (collective entered by one rank only?)")
        outs.append((p.returncode, o, e))
    for rc, o, e in outs:
        assert rc == 0, e[-2000:]
    assert outs[0][1].strip().endswith("done") and "done" not in outs[1][1]
    name = "ae_2048_attn" if script == "ae_v2" else None
    found = [os.path.join(d, f) for d, _, fs in os.walk(tmp_path) for f in fs if f == "last.ckpt"]
    assert len(found) == 1, found
    ck = torch.load(found[0], map_location="cpu", weights_only=False)
    assert ck["global_step"] == 2 and len(ck["optimizer_states"]) == (1 if script == "ae_v2" else 2)
    steps = [json.loads(l) for l in outs[0][1].splitlines() if l.startswith("{") and '"step"' in l]
    assert steps, "rank 0 logged no step"


def test_ssim_loss_branch_full_step_matches_oracle(dev):
    """The optional `1 - SSIM` term of the reference Loss (experiments/ae_v2/train.py:57-63, active when
    lpips.perceptual_weight > 0) END TO END: one full 128x128 training step through experiments/ae_v2/train.py::Loss
    (SsimFn inside autograd, both loss terms feeding one backward pass) against oracle.train_step(perceptual_weight=0.5)
    on the same weights and frames.  "Parity unpinned" for the SSIM part: pytorch_msssim is not installed and the
    reference holds no fixture for it — the oracle restates its published defaults."""
    import numpy as np
    from oracle import ae_oracle as orc
    from weatherforecastingtoolkit_amd import config as C, synth
    from weatherforecastingtoolkit_amd.experiments.ae_v2 import train
    cfg = C.load(os.path.join(os.path.dirname(train.__file__), "config.yaml"), train.CARRIED_KEYS)
    cfg.trainer.total_train_steps = 10
    cfg.lpips.disc_start = 10
    cfg.lpips.perceptual_weight = 0.5
    model = train.Model(cfg, img_size=128, variant="lin").to(dev).train()
    assert model.loss.perceptual_weight == 0.5
    np_sd = synth.synth_state_dict(synth.ae_state_dict_spec(128), seed=0)
    model.autoencoder.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in np_sd.items()}, strict=True)
    model.configure_optimizers()
    x = torch.from_numpy(synth.uniform_frames(2, 128, seed=1234))
    loss, logs = model.training_step({"vil": x.to(dev)}, 1)
    loss.backward()
    osd = orc.to_torch_sd(np_sd)
    _, _, oloss = orc.train_step(x, osd, None, perceptual_weight=0.5)
    _, _, oloss_l1 = orc.train_step(x, orc.to_torch_sd(np_sd), None, perceptual_weight=0.0)
    assert oloss > oloss_l1 * 1.05                      # the SSIM term is really in the loss
    assert abs(loss.item() - oloss) < 1e-5 * abs(oloss), (loss.item(), oloss)
    assert abs(float(logs["train/rec_loss"]) - oloss) < 1e-5 * abs(oloss)
    ref = dict(orc.trainable(osd))
    worst = 0.0
    for n, p in model.autoencoder.named_parameters():
        a, b = p.grad.double().norm().item(), ref[n].grad.double().norm().item()
        worst = max(worst, abs(a - b) / (b + 1e-12))
    assert worst < 5e-4, worst
