"""GAN half of the reference's `Loss` modules (experiments/ae_v2/train.py:27-102 and
experiments/ae_v2_2/train.py:29-95, identical arithmetic): PatchGAN discriminator, adaptive weight,
generator term and hinge discriminator loss, on libwfae.so kernels."""
from __future__ import annotations

import torch
import torch.nn as tnn

from .. import functional as Fn
from .. import ops
from ..pipeline.models.autoencoderkl.losses import NLayerDiscriminator, hinge_d_loss, weights_init


class frozen:
    """requires_grad=False on `params` inside the block — what Lightning's toggle_optimizer does to the
    other optimiser's parameters (ae_v2_2/train.py:133,145): no weight-gradient kernels are launched
    for them."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]

    def __enter__(self):
        for p in self.params:
            p.requires_grad_(False)

    def __exit__(self, *exc):
        for p in self.params:
            p.requires_grad_(True)


class GanLoss(tnn.Module):
    def __init__(self, disc_start, disc_num_layers=3, disc_in_channels=1, disc_weight=1.0, use_actnorm=False):
        super().__init__()
        self.disc_start, self.disc_weight = disc_start, disc_weight
        self.discriminator = NLayerDiscriminator(input_nc=disc_in_channels, n_layers=disc_num_layers,
                                                 use_actnorm=use_actnorm).apply(weights_init)

    def calculate_adaptive_weight(self, rec_loss, disc_loss, last_layer):
        """clamp(disc_weight * ||d rec / d w_last|| / (||d g / d w_last|| + 1e-4), 0, 1e4), detached
        (ae_v2_2/train.py:46-52).  Both probes write into fresh buffers (not the gradient arena) and
        only their sums of squares leave the device kernels."""
        with Fn.detached_grads():
            rec_grad = torch.autograd.grad(rec_loss, last_layer, retain_graph=True)[0]
            disc_grad = torch.autograd.grad(disc_loss, last_layer, retain_graph=True)[0]
        Fn.join_side_stream()
        return ops.adaptive_weight(ops.sumsq(rec_grad.contiguous().view(-1)), ops.sumsq(disc_grad.contiguous().view(-1)),
                                   self.disc_weight)

    def generator_loss(self, rec_loss, reconstructions, last_layer):
        """rec + d_weight * (-mean(D(x_hat)))  (ae_v2_2/train.py:69-79)"""
        g_loss = Fn.neg_mean(self.discriminator(reconstructions))
        if torch.is_grad_enabled() and rec_loss.requires_grad and last_layer is not None:
            d_weight = self.calculate_adaptive_weight(rec_loss, g_loss, last_layer)
        else:  # validation: the reference catches the RuntimeError and uses 0 (:74-76)
            d_weight = torch.zeros((), device=rec_loss.device)
        loss = Fn.AddFn.apply(rec_loss, Fn.ScaleByFn.apply(g_loss, d_weight))
        return loss, g_loss, d_weight

    def discriminator_loss(self, inputs, reconstructions):
        """hinge loss on D(x) and D(x_hat.detach())  (ae_v2_2/train.py:87-90)"""
        logits_real = self.discriminator(inputs.detach())
        logits_fake = self.discriminator(reconstructions.detach())
        return hinge_d_loss(logits_real, logits_fake), logits_real, logits_fake
