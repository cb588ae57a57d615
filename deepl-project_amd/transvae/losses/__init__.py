from .vae_loss import TransVAELoss, fused_l1_kl

__all__ = ["TransVAELoss", "fused_l1_kl"]
