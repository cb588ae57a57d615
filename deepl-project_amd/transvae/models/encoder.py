"""TransVAE encoder on the HIP path (interface mirror of R/transvae/models/encoder.py:31-126).

conv_in -> [ResBlocks]x2 stages -> [TransVAEBlocks] for the remaining stages, a Downsample
between stages.  Activations stay bf16 NHWC from the stem to the last block; the stem reads the
NCHW fp32 image directly (im2col to K=32, then one GEMM).
"""
from __future__ import annotations

from typing import List

import torch
import torch.nn as nn
import torch.nn.functional as F
import torch.utils.checkpoint

from ..hip import ops
from ..modules.blocks import ResBlock, TransVAEBlock
from ..modules.upsample import Downsample


def _tap(taps, key, h):
    if taps is not None:
        taps[key] = h.detach()
        sub = taps.get("@" + key)
        if sub is not None:
            h = sub(h)
    return h


def _checkpointed(block, h, scope="all"):
    """The activation-checkpointing toggle (R/transvae/models/encoder.py:97-99,117-118; decoder.py:98-100,118-119).
    ResBlocks (the stages that hold the large tensors) recompute only their two GroupNorm+SiLU outputs in the backward pass
    (fused-op recompute: no convolution runs twice, 2 saved full-resolution tensors per block instead of 4); TransVAE
    blocks are re-run as a whole like the reference does."""
    if getattr(block, "supports_fused_recompute", False) and isinstance(block.shortcut, nn.Identity):
        return block.forward_nhwc(h, recompute=True)
    if scope == "resblocks":
        return block.forward_nhwc(h)
    return torch.utils.checkpoint.checkpoint(block.forward_nhwc, h, use_reentrant=False)


class TransVAEEncoder(nn.Module):
    NUM_CNN_STAGES = 2  # hard-coded in the reference (encoder.py:60)

    def __init__(self, input_channels: int = 3, latent_dim: int = 32, depths: List[int] = (3, 3, 3, 4, 6),
                 base_dims: List[int] = (192, 192, 384, 768, 1536), compression_ratio: int = 16, mlp_ratio: float = 1.0,
                 head_dim: int = 64, use_rope: bool = True, use_conv_ffn: bool = True, use_dc_path: bool = True):
        super().__init__()
        depths, base_dims = list(depths), list(base_dims)
        self.num_stages = len(depths)
        self.depths, self.base_dims, self.compression_ratio = depths, base_dims, compression_ratio
        self.input_channels = input_channels
        self.conv_in = nn.Conv2d(input_channels, base_dims[0], 3, padding=1)
        self.stages = nn.ModuleList()
        self.downsamples = nn.ModuleList()
        for i, (depth, dim) in enumerate(zip(depths, base_dims)):
            if i < self.NUM_CNN_STAGES:
                blocks = [ResBlock(dim, dim) for _ in range(depth)]
            else:
                blocks = [TransVAEBlock(dim=dim, mlp_ratio=mlp_ratio, head_dim=head_dim, use_rope=use_rope,
                                        use_conv_ffn=use_conv_ffn) for _ in range(depth)]
            self.stages.append(nn.ModuleList(blocks))
            if i < self.num_stages - 1:
                self.downsamples.append(Downsample(dim, base_dims[i + 1], use_dc_path=use_dc_path))
        self.gradient_checkpointing = False

    def enable_gradient_checkpointing(self, scope: str = "all"):
        """scope "all": every block, like the reference's toggle; "resblocks": only the CNN stages' fused-op recompute (the
        stages that hold the large tensors; no convolution runs twice)."""
        if scope not in ("all", "resblocks"):
            raise ValueError(f"unknown checkpointing scope {scope!r}")
        self.gradient_checkpointing = scope

    def _stem(self, x: torch.Tensor) -> torch.Tensor:
        B, C, H, W = x.shape
        k = 9 * C
        kpad = (k + 31) // 32 * 32
        cols = ops.im2col3x3(x, kpad)                                   # [B*H*W, kpad], (ky,kx,c) order
        w = F.pad(self.conv_in.weight.permute(0, 2, 3, 1).reshape(-1, k), (0, kpad - k))
        return ops.linear(cols, w, self.conv_in.bias).view(B, H, W, -1)

    @ops.hip_entry
    def forward_nhwc(self, x: torch.Tensor, taps=None) -> torch.Tensor:
        """x: [B, C, H, W] image (any float dtype) -> [B, H/f, W/f, C_last] bf16.
        taps: optional dict receiving every intermediate ([B,H,W,C] bf16) under the oracle's key names; an entry
        '@' + key (callable) replaces that intermediate (precision attribution, tests/precision_report.py)."""
        f = 2 ** (self.num_stages - 1)
        if x.shape[-1] % f or x.shape[-2] % f:
            raise RuntimeError(f"input {tuple(x.shape[-2:])} must be divisible by {f}")
        h = _tap(taps, "encoder.conv_in", self._stem(x))
        for i, stage in enumerate(self.stages):
            for j, block in enumerate(stage):
                if self.gradient_checkpointing and self.training:
                    h = _checkpointed(block, h, self.gradient_checkpointing)
                else:
                    h = block.forward_nhwc(h)
                h = _tap(taps, f"encoder.stages.{i}.{j}", h)
            if i < len(self.downsamples):
                h = _tap(taps, f"encoder.downsamples.{i}", self.downsamples[i].forward_nhwc(h))
        return h

    @ops.hip_entry
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        h = self.forward_nhwc(x)
        return ops.to_nchw(h, 0, h.shape[-1])
