"""dec[2].up of the golden B=4 384x384 step: is the input gradient the unit produces INSIDE the model step the one the
same unit produces on the same tensors in isolation, and the one torch computes on the CPU?"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests._util import golden, relerr  # noqa: E402
from tests.test_model_gpu import _build, _frames  # noqa: E402
from weatherforecastingtoolkit_amd import functional as Fn, nn as wnn  # noqa: E402

gname = sys.argv[1] if len(sys.argv) > 1 else "g4_full384_b4"
g = golden(gname)
dev = torch.device("cuda:0")
net = _build(int(g["img_size"]), dev)
x = _frames(g).to(dev)
rec = {}
orig = Fn.UpUnitFn.backward


def patched(ctx, da):
    out = orig(ctx, da)
    xs, w = ctx.saved_tensors[0], ctx.saved_tensors[1]
    if tuple(w.shape[:2]) == (1024, 512):
        rec.update(da=da.detach().clone(), x=xs.detach().clone(), w=w.detach().clone(), gamma=ctx.saved_tensors[2].detach().clone(),
                   beta=ctx.beta.detach().clone(), dx=out[0].detach().clone(), dw=out[1].detach().clone())
    return out


Fn.UpUnitFn.backward = staticmethod(patched)
recon, z = net(x)
Fn.l1_loss(recon, x).backward()
torch.cuda.synchronize()
Fn.UpUnitFn.backward = staticmethod(orig)
print("captured", {k: tuple(v.shape) for k, v in rec.items()})


def unit(xin, w, gamma, beta, da):
    bn = wnn.BatchNorm2d(w.shape[1]).to(dev).train()
    with torch.no_grad():
        bn.weight.copy_(gamma)
        bn.bias.copy_(beta)
    xd, wd = xin.clone().requires_grad_(True), w.clone().requires_grad_(True)
    y = Fn.UpUnitFn.apply(xd, wd, bn.weight, bn.bias, bn)
    y.backward(da)
    torch.cuda.synchronize()
    return xd.grad, wd.grad


dx2, dw2 = unit(rec["x"], rec["w"], rec["gamma"], rec["beta"], rec["da"])
nb = rec["x"].shape[0]
print("in-model vs isolated rerun: dx equal", bool(torch.equal(dx2, rec["dx"])), " dw equal", bool(torch.equal(dw2, rec["dw"])))
for n in range(nb):
    print(f"  image {n}: |dx_model - dx_rerun| / |dx_rerun| = {float((rec['dx'][n] - dx2[n]).norm() / dx2[n].norm()):.3e}")
# truth: torch CPU fp32 (16 threads)
torch.set_num_threads(16)
xr, wr = rec["x"].cpu().requires_grad_(True), rec["w"].cpu().requires_grad_(True)
gr, br = rec["gamma"].cpu().requires_grad_(True), rec["beta"].cpu().requires_grad_(True)
y = F.gelu(F.batch_norm(F.conv_transpose2d(xr, wr, stride=2, padding=1), None, None, gr, br, True, 0.1, 1e-5))
y.backward(rec["da"].cpu())
for n in range(nb):
    print(f"  image {n}: model vs CPU {float((rec['dx'][n].cpu() - xr.grad[n]).norm() / xr.grad[n].norm()):.3e}   "
          f"rerun vs CPU {float((dx2[n].cpu() - xr.grad[n]).norm() / xr.grad[n].norm()):.3e}")
print("dw: model vs CPU", relerr(rec["dw"], wr.grad), " rerun vs CPU", relerr(dw2, wr.grad))
# per-image rerun (B = 1 slices use per-image batch statistics: not comparable) -> instead the direct (non-Winograd) form
from weatherforecastingtoolkit_amd import ops  # noqa: E402
ops.set_winograd(False)
dx3, dw3 = unit(rec["x"], rec["w"], rec["gamma"], rec["beta"], rec["da"])
for n in range(nb):
    print(f"  image {n}: direct-form rerun vs CPU {float((dx3[n].cpu() - xr.grad[n]).norm() / xr.grad[n].norm()):.3e}")
