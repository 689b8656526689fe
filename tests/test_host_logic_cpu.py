"""CPU: host-side logic — state_dict contract, config surface, loader indexing,
flat-arena bookkeeping."""
import numpy as np
import pytest
import torch

from tests._util import golden
from weatherforecastingtoolkit_amd import config as C
from weatherforecastingtoolkit_amd import synth
from weatherforecastingtoolkit_amd.pipeline import helpers
from weatherforecastingtoolkit_amd.pipeline.models.ae_64x8x8_lin import PosAwareAE_TF, Bottleneck, EncBlock, DecBlock


@pytest.mark.parametrize("size,nparams", [(128, 80750017), (384, 215033281)])
def test_state_dict_contract(size, nparams):
    """keys / shapes / dtypes / parameter order of SURVEY.md Appendix A (measured on the reference)."""
    net = PosAwareAE_TF(img_size=size)
    spec = synth.ae_state_dict_spec(size)
    sd = net.state_dict()
    assert list(sd.keys()) == [k for k, _, _ in spec] and len(sd) == 635
    for k, shp, kind in spec:
        assert tuple(sd[k].shape) == tuple(shp), k
        assert sd[k].dtype == (torch.int64 if kind == "bn_n" else torch.float32), k
    assert sum(p.numel() for p in net.parameters()) == nparams
    g = golden("g3_full128_b2")
    assert [n for n, _ in net.named_parameters()] == [str(n) for n in g["grad_names"]]
    assert sd["dec.1.up.0.weight"].shape == (1024, 1024, 4, 4) and sd["dec.4.up.0.weight"].shape == (256, 128, 4, 4)


def test_ctor_signatures_and_members():
    net = PosAwareAE_TF(1, 64, 8, 2048)
    for m in ("enc", "dec", "pos_emb", "to_latent", "from_latent", "act", "latent_channels"):
        assert hasattr(net, m)
    assert net.dec[-1].weight.shape == (1, 128, 3, 3)          # get_last_layer, reference train.py:207
    assert isinstance(net.enc[0], EncBlock) and isinstance(net.dec[1], DecBlock)
    b = Bottleneck(128, groups=8)
    assert b.f[5].groups == 8 and b.f[2].weight.shape == (32, 128, 1, 1)
    assert Bottleneck(16, groups=8).f[5].groups == 4            # g = min(groups, mid)
    # loading a Lightning-style checkpoint dict: 'autoencoder.' prefix stripped by the caller
    sd = {k: v for k, v in net.state_dict().items()}
    PosAwareAE_TF().load_state_dict(sd, strict=True)


def test_bn_counter_materialised_in_state_dict():
    from weatherforecastingtoolkit_amd.nn import BatchNorm2d
    bn = BatchNorm2d(4)
    bn._nbt_pending = 3
    assert int(bn.state_dict()["num_batches_tracked"]) == 3 and bn._nbt_pending == 0


def test_config_surface():
    import os
    import weatherforecastingtoolkit_amd.experiments.ae_v2 as pkg
    from weatherforecastingtoolkit_amd.experiments.ae_v2.train import CARRIED_KEYS
    cfg = C.load(os.path.join(os.path.dirname(pkg.__file__), "config.yaml"), CARRIED_KEYS)
    need = {"lpips": ["disc_start", "disc_weight", "disc_beta1", "disc_beta2", "disc_start_lr", "disc_peak_lr",
                      "disc_final_lr", "disc_warmup_ratio", "disc_in_channels", "disc_num_layers", "use_actnorm",
                      "perceptual_weight", "kl_weight", "logvar_init", "recon_weight"],
            "dataset": ["name", "seq_len", "stride", "batch_size", "num_workers", "input_frames", "pred_frames",
                        "image_width", "image_height", "channels"],
            "optim": ["lr", "weight_decay", "beta1", "beta2", "gradient_clip_val"],
            "cosine_warmup": ["start_lr", "peak_lr", "final_lr", "warmup_ratio"],
            "one_cycle": ["peak_lr", "start_lr", "final_lr", "rampup_ratio"],
            "lr_range_test": ["max_lr", "num_iter"],
            "trainer": ["devices", "max_epochs", "accumulate_grad_batches", "total_train_steps", "total_val_steps",
                        "total_test_steps", "save_every_n_steps", "save_on_train_epoch_end", "limit_train_batches",
                        "limit_val_batches", "limit_test_batches", "log_every_n_steps"],
            "logging": ["wandb_watch_log_freq", "log_train_all_metrics_n", "log_train_plots_n", "log_val_plots_n"]}
    for sec, keys in need.items():
        for k in keys:
            assert k in cfg[sec], (sec, k)
    for k in ("project_name", "experiment_path", "experiment_name"):
        assert k in cfg
    assert cfg.optim.lr == 5e-5 and cfg.cosine_warmup.peak_lr == 5e-5 and cfg.lpips.disc_start == 1.0
    cli = C.from_dotlist(["optim.lr=1e-4", "dataset.batch_size=32"])
    helpers.check_yaml(cfg, cli)
    assert C.merge(cfg, cli).dataset.batch_size == 32
    with pytest.raises(KeyError):
        helpers.check_yaml(cfg, C.from_dotlist(["optim.nope=1"]))


def test_loader_indexing_contract():
    """batch index -> (event, seq) pairs, length and u8 slices per reference sevire/sevir.py:979-1036."""
    from weatherforecastingtoolkit_amd.pipeline.datasets.sevire.sevir import SEVIRFrameLoader
    ev = (np.arange(3 * 4 * 5 * 7) % 251).astype(np.uint8).reshape(3, 4, 5, 7)
    ld = SEVIRFrameLoader(ev, batch_size=4, seq_len=2, stride=2)
    assert ld.num_seq_per_event == 1 + (7 - 2) // 2 == 3 and ld.total_num_seq == 9 and len(ld) == 2
    assert ld.sample_indices(0) == [(0, 0), (0, 1), (0, 2), (1, 0)]
    assert ld.sample_indices(1) == [(1, 1), (1, 2), (2, 0), (2, 1)]
    b = ld.batch_u8(1)
    assert b.shape == (4, 4, 5, 2) and np.array_equal(b[0], ev[1, :, :, 2:4]) and np.array_equal(b[3], ev[2, :, :, 2:4])
    # frame mode of ae_v2 (seq_len=1, stride=1): B temporally consecutive frames
    ld1 = SEVIRFrameLoader(ev, batch_size=5)
    assert [s for _, s in ld1.sample_indices(0)] == [0, 1, 2, 3, 4]
    # rank-strided sharding for data parallelism
    a, c = SEVIRFrameLoader(ev, 2, num_shard=2, rank=0), SEVIRFrameLoader(ev, 2, num_shard=2, rank=1)
    assert np.array_equal(a.batch_u8(1), SEVIRFrameLoader(ev, 2).batch_u8(2))
    assert np.array_equal(c.batch_u8(1), SEVIRFrameLoader(ev, 2).batch_u8(3))
    with pytest.raises(RuntimeError):
        ld[0]  # preprocessing runs on the GPU; no CPU fallback


def test_synth_is_reproducible_and_shaped():
    a, b = synth.uniform_frames(2, 16, seed=5), synth.uniform_frames(2, 16, seed=5)
    assert np.array_equal(a, b) and a.min() >= 0 and a.max() <= 1 and a.shape == (2, 1, 16, 16)
    ev = synth.blob_events(1, 32, 3)
    assert ev.dtype == np.uint8 and ev.shape == (1, 32, 32, 3) and (ev == 0).mean() > 0.2
    w = synth.synth_tensor(0, "enc.1.down.0.weight", (512, 256, 4, 4), "conv")
    assert abs(np.abs(w).max() - 1 / np.sqrt(256 * 16)) < 1e-4


def test_winograd_matrices_in_wino_hip_are_exact():
    """The F(2x2,2x2) and F(4x4,2x2) matrices compiled into csrc/wino.hip (parsed from the source) satisfy
    y = A^T [(G g) * (B^T d)] for the 2-tap correlation, in 1-D and 2-D, and In^T / Out^T are their transposes by
    construction.  F(4,2) is the Toom-Cook algorithm on the points {0, 1, -1, 2, -2}."""
    import os
    import re
    import numpy as np
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "weatherforecastingtoolkit_amd",
                            "csrc", "wino.hip")).read()

    def mat(struct, name):
        body = re.search(r"struct %s \{(.*?)\n\};" % struct, src, re.S).group(1)
        m = re.search(r"%s\[(\d+)\]\[(\d+)\] = \{(.*?)\};" % name, body, re.S)
        vals = re.findall(r"-?[\d.]+f?(?:\s*/\s*\d+)?", m.group(3))
        nums = []
        for v in vals:
            v = v.replace("f", "")
            nums.append(eval(v))
        return np.array(nums, dtype=np.float64).reshape(int(m.group(1)), int(m.group(2)))

    rng = np.random.default_rng(0)
    for struct, n, m in (("W22", 3, 2), ("W42", 5, 4)):
        BT, G, AT = mat(struct, "BT"), mat(struct, "G"), mat(struct, "AT")
        assert BT.shape == (n, n) and G.shape == (n, 2) and AT.shape == (m, n)
        d, g = rng.standard_normal(n), rng.standard_normal(2)
        y = AT @ ((G @ g) * (BT @ d))
        ref = np.array([d[o] * g[0] + d[o + 1] * g[1] for o in range(m)])
        assert np.abs(y - ref).max() < 1e-12
        d2, g2 = rng.standard_normal((n, n)), rng.standard_normal((2, 2))
        y2 = AT @ ((G @ g2 @ G.T) * (BT @ d2 @ BT.T)) @ AT.T
        ref2 = np.array([[sum(d2[o + a, p + b] * g2[a, b] for a in range(2) for b in range(2)) for p in range(m)] for o in range(m)])
        assert np.abs(y2 - ref2).max() < 1e-12


def test_one_cycle_closed_form_matches_torch():
    """helpers.one_cycle_scheduler (reference pipeline/helpers.py:109-140) against torch's OneCycleLR, lr and beta1"""
    import torch
    from weatherforecastingtoolkit_amd.optim import OneCycleLR

    class _Opt:            # the closed form only touches param_groups
        def __init__(self):
            self.param_groups = [{"lr": 1.0, "betas": (0.9, 0.999)}]

    for total, ramp, start, peak, final in [(100, 30, 4e-5, 1e-3, 4e-7), (37, 9.25, 1e-5, 2e-4, 1e-8), (10, 5, 1e-4, 1e-3, 1e-6)]:
        p = [torch.nn.Parameter(torch.zeros(1))]
        topt = torch.optim.AdamW(p, lr=1e-3)
        tsch = torch.optim.lr_scheduler.OneCycleLR(topt, max_lr=peak, total_steps=total, pct_start=ramp / total,
                                                   div_factor=peak / start, final_div_factor=start / final,
                                                   anneal_strategy="cos")
        o = _Opt()
        s = OneCycleLR(o, peak, total, ramp / total, peak / start, start / final)
        for step in range(total):
            assert abs(o.param_groups[0]["lr"] - topt.param_groups[0]["lr"]) <= 1e-12 * peak, step
            assert abs(o.param_groups[0]["betas"][0] - topt.param_groups[0]["betas"][0]) <= 1e-12, step
            if step + 1 < total:
                topt.step()
                tsch.step()
                s.step()
