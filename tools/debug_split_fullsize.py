"""split vs fp32-MFMA Winograd products at the B = 32, 384x384 layer shapes: max |diff| / rms per product"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from weatherforecastingtoolkit_amd import ops
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
torch.manual_seed(0)
for chi, clo, hlo in [(256, 512, 96), (512, 1024, 48), (1024, 1024, 24), (128, 256, 192)]:
    hi = torch.rand(B, chi, 2 * hlo, 2 * hlo, device=dev) - 0.5
    lo = (torch.rand(B, clo, hlo, hlo, device=dev) - 0.5) * 1e-4
    w = (torch.rand(clo, chi, 4, 4, device=dev) - 0.5) * 0.05
    res = {}
    for split in (False, True):
        ops.set_split_gemm(split)
        dw = torch.empty_like(w)
        ops.conv4x4s2_wgrad(lo, hi, dw)
        res[split] = (ops.conv4x4s2_down(hi, w), ops.conv4x4s2_up(lo, w), dw)
    for nm, a, b in zip(("down", "up", "wgrad"), res[False], res[True]):
        d = (a - b).abs()
        rms = a.pow(2).mean().sqrt()
        i = int(d.argmax())
        print(f"hi {chi}@{2*hlo} lo {clo}@{hlo} {nm:5s}: max|diff|/rms {float(d.max() / rms):.2e}  mean|diff|/rms {float(d.mean() / rms):.2e}  "
              f"norm ratio {float(b.double().norm() / a.double().norm()):.8f}  argmax {i} of {a.numel()}", flush=True)
    del hi, lo, w, res
