"""CPU: the oracle (oracle/ae_oracle.py) against the golden vectors produced by the
REAL reference (tests/golden/make_goldens.py).  The oracle uses the same torch CPU
built-ins the reference executes, so agreement is bit-exact."""
import numpy as np
import torch

from oracle import ae_oracle as orc
from tests._util import golden, relerr
from weatherforecastingtoolkit_amd import synth


def _block_sd(g, name):
    sd = {}
    for k in g.files:
        if k.startswith(name + "/sd/"):
            v = torch.from_numpy(g[k])
            if v.dtype.is_floating_point and "running_" not in k:
                v.requires_grad_(True)
            sd["m." + k[len(name) + 4:]] = v
    return sd


def test_blocks_bit_exact():
    g = golden("g2_blocks")
    for name, fn in [("bottleneck32", orc.bottleneck), ("bottleneck64", orc.bottleneck),
                     ("encblock", orc.enc_block), ("encblock2", orc.enc_block), ("decblock", orc.dec_block)]:
        sd = _block_sd(g, name)
        x = torch.from_numpy(g[name + "/x"]).requires_grad_(True)
        y = fn(x, sd, "m", True)
        y.backward(torch.from_numpy(g[name + "/gy"]))
        assert np.array_equal(y.detach().numpy(), g[name + "/y"]), name
        assert np.array_equal(x.grad.numpy(), g[name + "/gx"]), name
        for k, v in sd.items():
            if v.requires_grad:
                assert np.array_equal(v.grad.numpy(), g[f"{name}/grad/{k[2:]}"]), (name, k)
            elif "running_" in k or "num_batches" in k:
                assert np.array_equal(v.numpy(), g[f"{name}/after/{k[2:]}"]), (name, k)


def _frames(g):
    size, batch = int(g["img_size"]), int(g["batch"])
    if str(g["frames"]) == "uniform":
        return torch.from_numpy(synth.uniform_frames(batch, size, seed=1234))
    ev = synth.blob_events(1, size, batch, seed=1234)
    return torch.from_numpy(ev[0].transpose(2, 0, 1)[:, None].astype(np.float32) * np.float32(1 / 255))


def _full(gname, steps=None):
    g = golden(gname)
    size = int(g["img_size"])
    torch.set_num_threads(8)
    sd = orc.to_torch_sd(synth.synth_state_dict(synth.ae_state_dict_spec(size), seed=0))
    x = _frames(g)
    s0, peak, fin, total, warm = g["sched"]
    opt = orc.make_optimizer([p for _, p in orc.trainable(sd)], lr=5e-5, weight_decay=1e-4)
    sch = orc.make_scheduler(opt, s0, fin, peak, total, warm)
    idx = torch.from_numpy(g["lattice"])
    n = int(g["steps"]) if steps is None else steps
    for s in range(n):
        recon, z, loss = orc.train_step(x, sd, opt, sch)
        if s == 0:
            assert np.array_equal(recon[:, 0][:, idx][:, :, idx].numpy(), g["recon_lattice"])
            assert np.array_equal(z.numpy(), g["z"])
            gn = np.array([p.grad.double().norm().item() for _, p in orc.trainable(sd)])
            assert np.allclose(gn, g["grad_norms"], rtol=1e-12)
        assert loss == float(g[f"loss{s}"])
        assert opt.param_groups[0]["lr"] == float(g[f"lr_after{s}"])
    return g, sd


def test_full_model_128_trajectory_bit_exact():
    g, sd = _full("g3_full128_b2")
    for k in ["enc.0.down.1", "enc.3.res.3.f.6", "dec.4.res.3.f.0"]:
        assert np.array_equal(sd[k + ".running_mean"].numpy(), g[f"after/{k}.running_mean"])
        assert int(sd[k + ".num_batches_tracked"]) == int(g[f"after/{k}.num_batches_tracked"])
    with torch.no_grad():
        er, ez = orc.forward(_frames(g), sd, training=False)
    assert np.array_equal(ez.numpy(), g["eval_z"])


def test_full_model_128_blobs_bit_exact():
    _full("g3_full128_b4_blobs")


def test_lr_schedule_closed_forms():
    """oracle.lr_at and the product's closed form vs the real torch SequentialLR (incl. non-integer warmup)."""
    from weatherforecastingtoolkit_amd.optim import cosine_warmup_lr
    g = golden("g8_sched")
    for i in range(4):
        s0, peak, fin, total, warm = g[f"{i}/cfg"]
        lrs = g[f"{i}/lrs"]
        a = np.array([orc.lr_at(e, s0, peak, fin, total, warm) for e in range(len(lrs))])
        b = np.array([cosine_warmup_lr(e, s0, peak, fin, total, warm) for e in range(len(lrs))])
        assert np.max(np.abs(a - lrs) / lrs) < 1e-12 and np.max(np.abs(b - lrs) / lrs) < 1e-12


def test_ssim_psnr_restatement_self_consistent():
    """UNPINNED third-party arithmetic: fp32 restatement vs the stored fp64 values, plus
    known-answer properties (SSIM(x,x)=1, symmetry)."""
    g = golden("g7_metrics")
    for i in range(4):
        t, p = torch.from_numpy(g[f"{i}/target"]), torch.from_numpy(g[f"{i}/pred"])
        assert abs(float(orc.ssim(p, t)) - float(g[f"{i}/ssim"])) < 1e-5
        assert abs(orc.psnr(p, t) - float(g[f"{i}/psnr"])) < 1e-3
        assert abs(float(orc.ssim(t, t)) - 1.0) < 1e-6
        assert abs(float(orc.ssim(p, t)) - float(orc.ssim(t, p))) < 1e-6


# ------------------------------------------------------------------ AE + GAN (G6) ---
def test_discriminator_oracle_bit_exact():
    """oracle discriminator vs the fixture produced by the reference NLayerDiscriminator (model.py:100-150)"""
    g = golden("g6_gan128_b2")
    dnp = synth.synth_state_dict(synth.disc_state_dict_spec(1, 64, 3), seed=5)
    for tag, x in (("a", synth.uniform_frames(2, 128, seed=77)),
                   ("a32", synth.uniform(int(g["a32/seed"]), "disc32/x", (2, 1, 32, 32), 0, 1))):
        sd = orc.to_torch_sd(dnp)
        xt = torch.from_numpy(x).requires_grad_(True)
        y = orc.disc_forward(xt, sd, True)
        y.backward(torch.from_numpy(g[f"{tag}/gy"]))
        assert np.array_equal(y.detach().numpy(), g[f"{tag}/y"])
        assert np.array_equal(xt.grad.numpy(), g[f"{tag}/gx"])
        for k, v in sd.items():
            if f"{tag}/grad/{k}" in g.files:
                assert np.array_equal(v.grad.numpy(), g[f"{tag}/grad/{k}"]), k
            elif f"{tag}/grad_head/{k}" in g.files:
                assert np.array_equal(v.grad.flatten()[:2048].numpy(), g[f"{tag}/grad_head/{k}"]), k
        if tag == "a":
            for k, v in sd.items():
                if "running_" in k or "num_batches" in k:
                    assert np.array_equal(v.numpy(), g[f"a/after/{k}"]), k
            with torch.no_grad():
                assert np.array_equal(orc.disc_forward(xt.detach(), sd, False).numpy(), g["a/eval_y"])


def test_gan_step_oracle_bit_exact():
    """two G-then-D steps of the oracle vs the fixture made with the reference AE + discriminator modules in
    the step order of experiments/ae_v2_2/train.py:126-159"""
    g = golden("g6_gan128_b2")
    lr, wd, total, warm, clip = [float(v) for v in g["b/cfg"]]
    sd = orc.to_torch_sd(synth.synth_state_dict(synth.ae_state_dict_spec(128), seed=0))
    dsd = orc.to_torch_sd(synth.synth_state_dict(synth.disc_state_dict_spec(1, 64, 3), seed=5))
    og = orc.make_optimizer([p for _, p in orc.trainable(sd)], lr=lr, weight_decay=wd)
    od = orc.make_optimizer([p for _, p in orc.trainable(dsd)], lr=lr, weight_decay=wd)
    ogs = orc.make_scheduler(og, 5e-6, 5e-7, 5e-5, int(total), warm)
    ods = orc.make_scheduler(od, 5e-6, 5e-7, 5e-5, int(total), warm)
    x = torch.from_numpy(synth.uniform_frames(2, 128, seed=1234))
    idx = g["b/lattice"]
    for s in range(2):
        recon, log = orc.gan_train_step(x, sd, dsd, og, od, ogs, ods, True, 1.0, 1.0, clip)
        assert np.array_equal(recon[:, 0][:, idx][:, :, idx].numpy(), g[f"b/recon_lattice{s}"])
        for k, v in log.items():
            assert v == float(g[f"b/{k}{s}"]), (s, k)
        an = np.array([p.detach().double().norm().item() for _, p in orc.trainable(sd)])
        dn = np.array([p.detach().double().norm().item() for _, p in orc.trainable(dsd)])
        assert np.array_equal(an, g[f"b/ae_param_norms{s}"]) and np.array_equal(dn, g[f"b/d_param_norms{s}"])


def test_linear_forecaster_oracle_matches_fixture():
    """G9 (unpinned fixture): the oracle restatement reproduces it bit-exactly and equals a direct loop form"""
    g = golden("g9_linear_forecast")
    for case in range(3):
        b, t, tin, c, h, w = [int(x) for x in g[f"{case}/cfg"]]
        v = torch.from_numpy(synth.uniform(9, f"lf{case}/v", (b, t, c, h, w), -1, 1))
        wt = torch.from_numpy(synth.uniform(9, f"lf{case}/w", ((t - tin) * c, tin * c), -0.1, 0.1)).requires_grad_(True)
        bs = torch.from_numpy(synth.uniform(9, f"lf{case}/b", ((t - tin) * c,), -0.1, 0.1)).requires_grad_(True)
        loss, pred_abs = orc.linear_forecast_loss(v, wt, bs, tin)
        loss.backward()
        assert loss.item() == float(g[f"{case}/loss"])
        assert np.array_equal(pred_abs.detach().numpy(), g[f"{case}/pred_abs"])
        assert np.array_equal(wt.grad.numpy(), g[f"{case}/gw"])
        # independent per-pixel loop form of the same definition (one latent pixel)
        last = v[0, tin - 1, :, 0, 0]
        x = (v[0, :tin, :, 0, 0] - last).reshape(-1)
        y = (wt.detach() @ x + bs.detach()).reshape(t - tin, c) + last
        assert torch.allclose(y, pred_abs[0, :, :, 0, 0].detach(), atol=1e-6)
