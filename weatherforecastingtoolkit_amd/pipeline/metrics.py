"""SSIM / PSNR of the reference's pipeline/metrics.py (:71-93) on gfx950 kernels.

Inputs are (b, t, c, h, w) in [0, 1] like the reference's calc_metrics; SSIM is
torchmetrics' StructuralSimilarityIndexMeasure(data_range=1.0) (11-tap Gaussian,
sigma 1.5 — numerically the valid-window form), PSNR is
per-sample PeakSignalNoiseRatio() with data_range = max(target) - min(target).
CRPS / CSI / HSS of the reference file are forecast-skill scores outside the AE
train step and are not built.
"""
from __future__ import annotations

from .. import ops


def _flat(x):
    b, t, c, h, w = x.shape
    return x.reshape(b * t * c, 1, h, w).contiguous()


def ssim(pred, target):
    """reference pipeline/metrics.py:71-75"""
    return float(ops.ssim_fwd(_flat(target.detach()), _flat(pred.detach()), clamp01=False).item())


def psnr(pred, target):
    """reference pipeline/metrics.py:77-84"""
    return float(ops.psnr(_flat(pred.detach()), _flat(target.detach()), clamp01=False).item())


def calc_metrics(pred, target):
    """reference pipeline/metrics.py:86-133 restricted to the AE path: clamp to [0,1]
    (:92-93, fused into the kernels) then SSIM and PSNR."""
    p, g = _flat(pred.detach()), _flat(target.detach())
    return {
        "paper_SSIM": float(ops.ssim_fwd(g, p, clamp01=True).item()),
        "paper_PSNR": float(ops.psnr(p, g, clamp01=True).item()),
    }
