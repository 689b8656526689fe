"""Loss helpers of `pipeline/models/autoencoderkl/losses/contperceptual.py` used by the AE+GAN step."""
from __future__ import annotations

from ..... import functional as Fn


def adopt_weight(weight, global_step, threshold=0, value=0.0):
    """reference contperceptual.py:13-16"""
    if global_step < threshold:
        weight = value
    return weight


def hinge_d_loss(logits_real, logits_fake):
    """0.5 * (mean(relu(1 - real)) + mean(relu(1 + fake)))  (reference contperceptual.py:19-23)"""
    return Fn.hinge_d_loss(logits_real, logits_fake)
