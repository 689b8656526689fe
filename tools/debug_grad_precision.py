"""Gradient accuracy vs an fp64 oracle: mine (GPU fp32) and the fp32 CPU oracle."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from weatherforecastingtoolkit_amd import synth, functional as Fn
from weatherforecastingtoolkit_amd.pipeline.models.ae_64x8x8_lin import PosAwareAE_TF
from oracle import ae_oracle as orc
dev = torch.device("cuda:0")
size, B = int(os.environ.get("SIZE", 128)), int(os.environ.get("B", 2))
np_sd = synth.synth_state_dict(synth.ae_state_dict_spec(size), seed=0)
x = torch.from_numpy(synth.uniform_frames(B, size, seed=1234))
net = PosAwareAE_TF(img_size=size).to(dev)
net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in np_sd.items()}, strict=True)
net.train()
xd = x.to(dev)
recon, z = net(xd); loss = Fn.l1_loss(recon, xd); loss.backward()
mine = {n: p.grad.detach().double().cpu() for n, p in net.named_parameters()}
sd32 = orc.to_torch_sd(np_sd)
r32, z32 = orc.forward(x, sd32, True); l32 = orc.loss_fn(r32, x); l32.backward()
g32 = {k: v.grad.double() for k, v in orc.trainable(sd32)}
sd64 = {}
for k, v in np_sd.items():
    t = torch.from_numpy(np.asarray(v).copy()) if v.ndim else torch.tensor(0)
    if t.dtype.is_floating_point:
        t = t.double()
        if "running_" not in k: t.requires_grad_(True)
    sd64[k] = t
r64, z64 = orc.forward(x.double(), sd64, True); l64 = orc.loss_fn(r64, x.double()); l64.backward()
print("loss mine %.10f  fp32 %.10f  fp64 %.10f" % (loss.item(), l32.item(), l64.item()))
print("recon err vs fp64: mine %.2e  fp32-oracle %.2e" % (float((recon.detach().double().cpu() - r64).abs().max()), float((r32.double() - r64).abs().max())))
rows = []
for k, v in sd64.items():
    if not v.requires_grad: continue
    g = v.grad; den = float(g.abs().max()) + 1e-300
    rows.append((float((mine[k] - g).abs().max()) / den, float((g32[k] - g).abs().max()) / den, k))
rows.sort(reverse=True)
print("worst (mine_err, fp32_oracle_err, name):")
for r in rows[:14]: print("  %.2e  %.2e  %s" % r)
me = np.array([r[0] for r in rows]); oe = np.array([r[1] for r in rows])
print("median mine %.2e oracle32 %.2e ; max mine %.2e oracle32 %.2e" % (np.median(me), np.median(oe), me.max(), oe.max()))
