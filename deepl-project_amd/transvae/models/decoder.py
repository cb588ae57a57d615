"""TransVAE decoder on the HIP path (interface mirror of R/transvae/models/decoder.py:29-132).

conv_in -> [TransVAEBlocks] stages -> last two stages of ResBlocks, an Upsample between stages,
then GroupNorm(32) - SiLU - conv_out.  Output is the unbounded reconstruction, NCHW fp32.
The 3-channel head is computed as a 32-wide GEMM tile (zero-padded output channels) and sliced at
the NCHW conversion; the latent input is zero-padded to 32 channels the same way.
"""
from __future__ import annotations

from typing import List

import torch
import torch.nn as nn
import torch.nn.functional as F
import torch.utils.checkpoint

from ..hip import ops
from ..modules.blocks import ResBlock, TransVAEBlock
from ..modules.upsample import Upsample
from .encoder import _checkpointed, _tap


def _round32(n: int) -> int:
    return (n + 31) // 32 * 32


class TransVAEDecoder(nn.Module):
    def __init__(self, latent_dim: int = 32, output_channels: int = 3, depths: List[int] = (6, 4, 3, 3, 3),
                 base_dims: List[int] = (1536, 768, 384, 192, 192), compression_ratio: int = 16, mlp_ratio: float = 1.0,
                 head_dim: int = 64, use_rope: bool = True, use_conv_ffn: bool = True, use_dc_path: bool = True):
        super().__init__()
        depths, base_dims = list(depths), list(base_dims)
        self.num_stages = len(depths)
        self.depths, self.base_dims = depths, base_dims
        self.latent_dim, self.output_channels = latent_dim, output_channels
        self.conv_in = nn.Conv2d(latent_dim, base_dims[0], 3, padding=1)
        self.stages = nn.ModuleList()
        self.upsamples = nn.ModuleList()
        n_tr = self.num_stages - 2
        for i, (depth, dim) in enumerate(zip(depths, base_dims)):
            if i < n_tr:
                blocks = [TransVAEBlock(dim=dim, mlp_ratio=mlp_ratio, head_dim=head_dim, use_rope=use_rope,
                                        use_conv_ffn=use_conv_ffn) for _ in range(depth)]
            else:
                blocks = [ResBlock(dim, dim) for _ in range(depth)]
            self.stages.append(nn.ModuleList(blocks))
            if i < self.num_stages - 1:
                self.upsamples.append(Upsample(dim, base_dims[i + 1], use_dc_path=use_dc_path))
        self.norm_out = nn.GroupNorm(32, base_dims[-1])
        self.conv_out = nn.Conv2d(base_dims[-1], output_channels, 3, padding=1)
        self.gradient_checkpointing = False

    def enable_gradient_checkpointing(self, scope: str = "all"):
        """scope "all": every block, like the reference's toggle; "resblocks": only the CNN stages' fused-op recompute (the
        stages that hold the large tensors; no convolution runs twice)."""
        if scope not in ("all", "resblocks"):
            raise ValueError(f"unknown checkpointing scope {scope!r}")
        self.gradient_checkpointing = scope

    @ops.hip_entry
    def forward_nhwc(self, z: torch.Tensor, taps=None) -> torch.Tensor:
        """z: [B, latent, h, w] NCHW (fp32) -> padded reconstruction [B, H, W, 32k] bf16.  taps: see the encoder."""
        L = self.latent_dim
        lp = _round32(L)
        zin = ops.to_nhwc(z, lp)
        w_in = F.pad(self.conv_in.weight.permute(0, 2, 3, 1), (0, lp - L))
        h = _tap(taps, "decoder.conv_in", ops.conv(zin, w_in, self.conv_in.bias, None, "c3s1"))
        for i, stage in enumerate(self.stages):
            for j, block in enumerate(stage):
                if self.gradient_checkpointing and self.training:
                    h = _checkpointed(block, h, self.gradient_checkpointing)
                else:
                    h = block.forward_nhwc(h)
                h = _tap(taps, f"decoder.stages.{i}.{j}", h)
            if i < len(self.upsamples):
                h = _tap(taps, f"decoder.upsamples.{i}", self.upsamples[i].forward_nhwc(h))
        h = ops.group_norm_silu(h, self.norm_out.weight, self.norm_out.bias, 32, self.norm_out.eps)
        co = self.output_channels
        cp = _round32(co)
        w_out = F.pad(self.conv_out.weight.permute(0, 2, 3, 1), (0, 0, 0, 0, 0, 0, 0, cp - co))
        b_out = F.pad(self.conv_out.bias, (0, cp - co))
        return ops.conv(h, w_out, b_out, None, "c3s1")

    @ops.hip_entry
    def forward(self, z: torch.Tensor) -> torch.Tensor:
        y = self.forward_nhwc(z)
        return ops.to_nchw(y, 0, self.output_channels)
