"""Drop-in for the pieces of `pipeline/models/autoencoderkl/losses/` the AE+GAN step uses
(SURVEY.md §8(f) next-2): PatchGAN discriminator, its initialiser and the hinge loss.
LPIPS (VGG16 weights fetched from the network, perceptual_weight 0.0 in every shipped config) is
out of scope."""
from .model import NLayerDiscriminator, weights_init  # noqa: F401
from .contperceptual import hinge_d_loss, adopt_weight  # noqa: F401
