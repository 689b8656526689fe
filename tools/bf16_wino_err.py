"""Relative L2 error of the 4x4 stride-2 convolution family under 'medium' (bf16 operand) precision, per algorithm:
direct implicit GEMM, Winograd F(2x2,2x2), Winograd F(4x4,2x2) — against the fp32 direct result on the same inputs.

    python tools/bf16_wino_err.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import weatherforecastingtoolkit_amd as pkg  # noqa: E402
from weatherforecastingtoolkit_amd import ops  # noqa: E402


def l2(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    for nb, chi, clo, hlo, wlo in [(4, 256, 512, 48, 48), (4, 512, 1024, 24, 24), (4, 128, 256, 96, 96)]:
        hi = torch.nn.functional.gelu(torch.randn((nb, chi, 2 * hlo, 2 * wlo), generator=g)).to(dev)   # post-GELU-like
        lo = (torch.randn((nb, clo, hlo, wlo), generator=g) * 1e-3).to(dev)                          # gradient-like
        w = (torch.randn((clo, chi, 4, 4), generator=g) * (chi * 16) ** -0.5).to(dev)

        def run():
            d = ops.conv4x4s2_down(hi, w)
            u = ops.conv4x4s2_up(lo, w)
            dw = torch.empty_like(w)
            ops.conv4x4s2_wgrad(lo, hi, dw)
            return d, u, dw

        pkg.set_float32_matmul_precision("highest")
        ops.set_winograd(False)
        ref = run()
        row = {}
        for prec in ("highest", "medium"):
            pkg.set_float32_matmul_precision(prec)
            for mode in (False, "f22", "f42"):
                ops.set_winograd(mode)
                got = run()
                row[(prec, str(mode))] = [l2(a, b) for a, b in zip(got, ref)]
        print(f"layer {chi}->{clo} @ {hlo}x{wlo} (down, up, wgrad rel-L2 vs fp32 direct)")
        for k, v in row.items():
            print(f"  {k[0]:8s} {k[1]:6s} " + "  ".join(f"{e:.2e}" for e in v))
    pkg.set_float32_matmul_precision("highest")
    ops.set_winograd("auto")


if __name__ == "__main__":
    main()
