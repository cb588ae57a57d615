"""transvae -- MI355X-native TransVAE forward/backward path.

Import surface of the reference package (R/transvae/__init__.py:5-9): `from transvae import
TransVAE, create_transvae, TransVAELoss`.  The loss re-exported here holds the closed-form L1 + KL terms
(one fused HIP pass, transvae/losses/vae_loss.py); the reference's LPIPS / VF / GAN terms need external networks
(`lpips` + VGG weights, DINOv2, a discriminator) and are out of scope.
"""
from .losses.vae_loss import TransVAELoss
from .models.transvae import TransVAE, create_transvae

__version__ = "0.2.0"
__all__ = ["TransVAE", "create_transvae", "TransVAELoss"]
