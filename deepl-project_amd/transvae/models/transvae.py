"""TransVAE on MI355X: the reference's nn.Module API over hand-written gfx950 kernels.

Drop-in for R/transvae/models/transvae.py: same constructor keywords, methods
(encode / reparameterize / decode / forward / get_last_layer / enable_gradient_checkpointing /
get_num_params / from_pretrained), public sub-modules (.encoder, .decoder, .conv_mu,
.conv_logvar) and state_dict keys, so `train.py`-style callers (TransVAE(config=model_cfg, ...),
DDP(model), model(images) -> (recon, mu, logvar)) and README-style callers
(TransVAE(variant='large', compression_ratio=16, latent_dim=32)) both work: `config` is optional
here and falls back to the reference's variant table (transvae.py:107-153, dead code there --
SURVEY F2/F3).

Inputs / outputs are NCHW fp32 like the reference; inside, activations are bf16 NHWC and every
op on them is a kernel from libtransvae_hip.so.  There is no CPU path: calling forward on CPU
tensors raises.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..hip import ops
from .decoder import TransVAEDecoder
from .encoder import TransVAEEncoder

VARIANT_CONFIGS = {
    "tiny_f16d32": {"depths": [3, 3, 3, 3, 3], "base_dims": [128, 128, 256, 256, 512]},
    "base_f16d32": {"depths": [3, 3, 3, 3, 3], "base_dims": [128, 128, 256, 512, 1024]},
    "large_f16d32": {"depths": [3, 3, 3, 4, 6], "base_dims": [192, 192, 384, 768, 1536]},
    "huge_f16d32": {"depths": [3, 3, 4, 6, 8], "base_dims": [256, 256, 512, 1024, 2048]},
    "giant_f16d32": {"depths": [3, 3, 4, 8, 10], "base_dims": [320, 320, 640, 1280, 2560]},
    "large_f8d16": {"depths": [3, 3, 6, 8], "base_dims": [192, 384, 768, 1536]},
}


def _round32(n: int) -> int:
    return (n + 31) // 32 * 32


class TransVAE(nn.Module):
    def __init__(self, config: Optional[dict] = None, variant: str = "large", compression_ratio: int = 16,
                 latent_dim: int = 32, input_channels: int = 3, use_rope: bool = True, use_conv_ffn: bool = True,
                 use_dc_path: bool = True, clamp_latent: bool = False, **kwargs):
        super().__init__()
        self.variant, self.compression_ratio, self.latent_dim = variant, compression_ratio, latent_dim
        self.clamp_latent = clamp_latent  # True = the patched copy's clamps (P/.../transvae.py:186-196,243-245)
        if config is None:
            config = self._get_variant_config(variant, compression_ratio, latent_dim)
        self.config = dict(config)
        depths, dims = list(config["depths"]), list(config["base_dims"])
        common = dict(compression_ratio=compression_ratio, mlp_ratio=config.get("mlp_ratio", 1.0),
                      head_dim=config.get("head_dim", 64), use_rope=use_rope, use_conv_ffn=use_conv_ffn,
                      use_dc_path=use_dc_path)
        self.encoder = TransVAEEncoder(input_channels=input_channels, latent_dim=latent_dim, depths=depths,
                                       base_dims=dims, **common)
        self.conv_mu = nn.Conv2d(dims[-1], latent_dim, 3, padding=1)
        self.conv_logvar = nn.Conv2d(dims[-1], latent_dim, 3, padding=1)
        self.decoder = TransVAEDecoder(latent_dim=latent_dim, output_channels=input_channels, depths=depths[::-1],
                                       base_dims=dims[::-1], **common)
        self._initialize_weights()
        # conv weights in channels_last memory = the [Cout,KH,KW,Cin] layout the kernels repack from
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last)

    @staticmethod
    def _get_variant_config(variant: str, f: int, d: int) -> dict:
        key = f"{variant}_f{f}d{d}"
        if key not in VARIANT_CONFIGS:
            raise ValueError(f"Unknown variant: {variant} with f{f}d{d}")
        cfg = dict(VARIANT_CONFIGS[key])
        cfg.update(mlp_ratio=1.0, head_dim=64)
        return cfg

    def _initialize_weights(self):
        """Same distributions as the reference (transvae.py:155-168)."""
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, (nn.LayerNorm, nn.GroupNorm)):
                nn.init.constant_(m.weight, 1.0)
                nn.init.constant_(m.bias, 0)

    # ------------------------------------------------------------------ path
    @ops.hip_entry
    def encode(self, x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """x [B,C,H,W] -> (mu, logvar), each [B, latent, H/f, W/f] fp32.  conv_mu and conv_logvar run
        as ONE conv with 2*latent output channels (transvae.py:182-183)."""
        h = self.encoder.forward_nhwc(x)
        L = self.latent_dim
        cp = _round32(2 * L)
        w = torch.cat([self.conv_mu.weight, self.conv_logvar.weight], 0).permute(0, 2, 3, 1)
        w = F.pad(w, (0, 0, 0, 0, 0, 0, 0, cp - 2 * L))
        b = F.pad(torch.cat([self.conv_mu.bias, self.conv_logvar.bias], 0), (0, cp - 2 * L))
        ml = ops.conv(h, w, b, None, "c3s1")
        return ops.to_nchw(ml, 0, L), ops.to_nchw(ml, L, L)

    def reparameterize(self, mu: torch.Tensor, logvar: torch.Tensor, eps: Optional[torch.Tensor] = None) -> torch.Tensor:
        """z = mu + eps * exp(logvar / 2); eps ~ N(0,1) from the global RNG unless supplied
        (transvae.py:186-199).  [B, latent, 16, 16] elementwise -- plain torch."""
        mu_f, lv = mu.float(), logvar.float()
        if self.clamp_latent:
            lv = lv.clamp(-30.0, 20.0)
        std = torch.exp(0.5 * lv)
        if eps is None:
            eps = torch.randn_like(std)
        return (mu_f + eps * std).to(mu.dtype)

    @ops.hip_entry
    def decode(self, z: torch.Tensor) -> torch.Tensor:
        return self.decoder(z)

    @ops.hip_entry
    def forward(self, x: torch.Tensor, return_dict: bool = False, eps: Optional[torch.Tensor] = None):
        mu, logvar = self.encode(x)
        if self.clamp_latent:
            mu = mu.clamp(-50, 50)
            logvar = logvar.clamp(-30, 20)
        z = self.reparameterize(mu, logvar, eps)
        reconstruction = self.decode(z)
        if return_dict:
            return {"reconstruction": reconstruction, "mu": mu, "logvar": logvar, "z": z}
        return reconstruction, mu, logvar

    # ------------------------------------------------------------------ helpers
    def get_last_layer(self):
        return self.decoder.conv_out.weight

    @classmethod
    def from_pretrained(cls, model_name: str, **kwargs):
        """'transvae-large-f16d32' -> constructor arguments; like the reference no weights are
        fetched (transvae.py:248-267: "weights TODO")."""
        variant, cfg = model_name.split("-")[1:3]
        f, d = int(cfg[1:].split("d")[0]), int(cfg.split("d")[1])
        return cls(variant=variant, compression_ratio=f, latent_dim=d, **kwargs)

    def enable_gradient_checkpointing(self, scope: str = "all"):
        """R/transvae/models/transvae.py:269-272.  scope (extension, default = the reference's behaviour): "resblocks" keeps
        the recompute to the CNN stages' GroupNorm+SiLU outputs (encoder.py: _checkpointed)."""
        self.encoder.enable_gradient_checkpointing(scope)
        self.decoder.enable_gradient_checkpointing(scope)

    def get_num_params(self) -> dict:
        enc = sum(p.numel() for p in self.encoder.parameters())
        dec = sum(p.numel() for p in self.decoder.parameters())
        return {"encoder": enc, "decoder": dec, "total": sum(p.numel() for p in self.parameters())}


def create_transvae(variant: str = "large", compression_ratio: int = 16, latent_dim: int = 32, **kwargs) -> TransVAE:
    return TransVAE(variant=variant, compression_ratio=compression_ratio, latent_dim=latent_dim, **kwargs)
