"""Is the gradient-norm spread of the split-operand Winograd GEMMs a precision effect or the L1 kink?
Runs the g4_full384_b1 frame (replicated) with split off / on, counts sign(recon - x) differences, and compares the
gradient norms (a) as computed, (b) with the backward pass driven by the SAME sign pattern (the fp32-MFMA run's)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from weatherforecastingtoolkit_amd import functional as Fn, ops
from tests.test_model_gpu import _build, _frames
from tests._util import golden

dev = torch.device("cuda:0")
g = golden("g4_full384_b1")
nrep = int(sys.argv[1]) if len(sys.argv) > 1 else 4
x = _frames(g).to(dev).expand(nrep, -1, -1, -1).contiguous()
out = {}
for split in (False, True):
    ops.set_split_gemm(split)
    net = _build(384, dev)
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.momentum = 0.0
    recon, z = net(x)
    d = (recon.detach() - x)
    out[split] = {"recon": recon.detach().clone(), "sign": torch.sign(d), "absd": d.abs()}
    Fn.l1_loss(recon, x).backward()
    Fn.join_side_stream()
    out[split]["gn"] = np.array([p.grad.double().norm().item() for p in net.parameters()])
    # the same network again, backward on the fp32 run's sign pattern
    net.zero_grad(set_to_none=True)
    recon, z = net(x)
    recon.backward(out[False]["sign"] / recon.numel())
    Fn.join_side_stream()
    out[split]["gn_fixed"] = np.array([p.grad.double().norm().item() for p in net.parameters()])
names = [n for n, _ in net.named_parameters()]
fl = out[False]["sign"] != out[True]["sign"]
print("recon relerr split vs fp32:", float((out[True]["recon"] - out[False]["recon"]).abs().max() / out[False]["recon"].abs().max()))
print("sign flips:", int(fl.sum()), "|recon-x| at flips:", out[False]["absd"][fl].tolist()[:8])
ref = g["grad_norms"]
for k in ("gn", "gn_fixed"):
    for split in (False, True):
        rel = np.abs(out[split][k] - ref) / (ref + 1e-12)
        print(f"{k:9s} split={split}: vs golden max {rel.max():.2e} ({names[int(rel.argmax())]}) median {np.median(rel):.2e}")
    rel = np.abs(out[True][k] - out[False][k]) / (out[False][k] + 1e-12)
    print(f"{k:9s} split vs fp32-MFMA: max {rel.max():.2e} ({names[int(rel.argmax())]}) median {np.median(rel):.2e}")
