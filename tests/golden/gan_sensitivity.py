"""How far does the REFERENCE arithmetic itself move when only rounding changes?  Runs two AE+GAN steps of
the oracle (bit-identical to the reference modules, see make_goldens.py g6) in fp32 and in fp64 on the G6
input and prints the relative differences.  LeakyReLU's slope jump at 0 and Adam's sign-like first step
amplify rounding: step 0 d_weight 7.8e-4, step 1 recon 3.0e-2 / d_weight 3.3e-2 / g_grad_norm 3.7e-2.
tests/test_gan_gpu.py takes its step-1 tolerances from this output.

    python tests/golden/gan_sensitivity.py
"""
import os
import sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import ae_oracle as orc
from weatherforecastingtoolkit_amd import synth
torch.set_num_threads(8)
np_sd = synth.synth_state_dict(synth.ae_state_dict_spec(128), seed=0)
dnp = synth.synth_state_dict(synth.disc_state_dict_spec(1, 64, 3), seed=5)
x = torch.from_numpy(synth.uniform_frames(2, 128, seed=1234))
out = {}
for dt in (torch.float32, torch.float64):
    def cast(np_d):
        sd = orc.to_torch_sd(np_d)
        return {k: (v.detach().to(dt).requires_grad_(v.requires_grad) if v.dtype.is_floating_point else v) for k, v in sd.items()}
    sd, dsd = cast(np_sd), cast(dnp)
    og = orc.make_optimizer([p for _, p in orc.trainable(sd)], lr=5e-5, weight_decay=1e-3)
    od = orc.make_optimizer([p for _, p in orc.trainable(dsd)], lr=5e-5, weight_decay=1e-3)
    ogs = orc.make_scheduler(og, 5e-6, 5e-7, 5e-5, 40, 4.0)
    ods = orc.make_scheduler(od, 5e-6, 5e-7, 5e-5, 40, 4.0)
    recs, logs = [], []
    for s in range(2):
        r, lg = orc.gan_train_step(x.to(dt), sd, dsd, og, od, ogs, ods, True, 1.0, 1.0, 1.0)
        recs.append(r.double()); logs.append(lg)
    out[dt] = (recs, logs)
a, b = out[torch.float32], out[torch.float64]
for s in range(2):
    d = (a[0][s] - b[0][s]).abs().max() / b[0][s].abs().max()
    print("step", s, "recon max-rel fp32 vs fp64:", d.item())
    for k in a[1][s]:
        print("   ", k, a[1][s][k], b[1][s][k], abs(a[1][s][k] - b[1][s][k]) / abs(b[1][s][k]))
