"""ResBlock / TransVAEBlock / RMSNorm on the HIP path.

Mirrors the interface of R/transvae/modules/blocks.py (class names, constructor arguments,
parameter names => identical state_dict keys) but computes on bf16 NHWC tensors with the gfx950
kernels in ``transvae.hip``.  The nn.Conv2d / nn.GroupNorm / nn.Linear children are parameter
containers only (they give the reference's keys, shapes and init); they are never called.

Every block exposes
    forward_nhwc(x)  x: [B, H, W, C] bf16 contiguous  (the fast path used by encoder / decoder)
    forward(x)       x: [B, C, H, W] like the reference (converted at the boundary)
"""
from __future__ import annotations

import torch
import torch.nn as nn

from ..hip import fused, ops
from .attention import FlashAttentionWithRoPE
from .conv import ConvFFN


def krsc(conv: nn.Conv2d) -> torch.Tensor:
    """OIHW master weight viewed as [Cout, KH, KW, Cin] (free when the parameter is channels_last)."""
    return conv.weight.permute(0, 2, 3, 1)


def nchw_call(mod, x: torch.Tensor) -> torch.Tensor:
    """Reference-style call on an NCHW tensor: convert in, run the NHWC path, convert out."""
    C = x.shape[1]
    y = mod.forward_nhwc(ops.to_nhwc(x, C))
    return ops.to_nchw(y, 0, y.shape[-1]).to(x.dtype if x.dtype.is_floating_point else torch.float32)


class ResBlock(nn.Module):
    """GN(32)-SiLU-conv3x3, GN(32)-SiLU-conv3x3, + shortcut  (R/transvae/modules/blocks.py:22-68)."""

    def __init__(self, in_channels: int, out_channels: int, use_conv_shortcut: bool = False):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.norm1 = nn.GroupNorm(32, in_channels)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, padding=1)
        self.norm2 = nn.GroupNorm(32, out_channels)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, padding=1)
        if in_channels != out_channels:
            k = 3 if use_conv_shortcut else 1
            self.shortcut = nn.Conv2d(in_channels, out_channels, k, padding=k // 2)
        else:
            self.shortcut = nn.Identity()

    # fused-op recompute available (the checkpointing toggle of the encoder / decoder uses it instead of re-running the block)
    supports_fused_recompute = True

    def forward_nhwc(self, x: torch.Tensor, recompute: bool = False) -> torch.Tensor:
        if isinstance(self.shortcut, nn.Identity):   # every shipped config: one fused forward/backward
            return fused.ResBlockFn.apply(x, self.norm1.weight, self.norm1.bias, krsc(self.conv1), self.conv1.bias,
                                          self.norm2.weight, self.norm2.bias, krsc(self.conv2), self.conv2.bias,
                                          self.norm1.eps, self.norm2.eps, recompute)
        a = ops.group_norm_silu(x, self.norm1.weight, self.norm1.bias, 32, self.norm1.eps)
        h = ops.conv(a, krsc(self.conv1), self.conv1.bias, None, "c3s1")
        a = ops.group_norm_silu(h, self.norm2.weight, self.norm2.bias, 32, self.norm2.eps)
        if isinstance(self.shortcut, nn.Identity):
            skip = x
        elif self.shortcut.kernel_size[0] == 3:
            skip = ops.conv(x, krsc(self.shortcut), self.shortcut.bias, None, "c3s1")
        else:
            B, H, W, C = x.shape
            w = self.shortcut.weight.view(self.out_channels, self.in_channels)
            skip = ops.linear(x.view(B * H * W, C), w, self.shortcut.bias).view(B, H, W, self.out_channels)
        return ops.conv(a, krsc(self.conv2), self.conv2.bias, skip, "c3s1")

    @ops.hip_entry
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return nchw_call(self, x)


class RMSNorm(nn.Module):
    """x / sqrt(mean_c(x^2) + eps) * weight   (R/transvae/modules/blocks.py:163-204).

    Inside TransVAEBlock only ``weight`` is used: the normalisation itself is fused with what
    follows it (LayerNorm-hat for attention, the proj_in GEMM for the FFN).  Standalone calls
    run the row-norm kernel and apply the weight afterwards.
    """

    def __init__(self, dim: int, eps: float = 1e-6):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim))

    def forward_tokens(self, t: torch.Tensor) -> torch.Tensor:  # [T, C] bf16
        return (ops.rms_hat(t, self.eps).float() * self.weight).to(torch.bfloat16)

    @ops.hip_entry
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.dim() == 4:
            B, C, H, W = x.shape
            t = ops.to_nhwc(x, C).view(B * H * W, C)
            y = self.forward_tokens(t).view(B, H, W, C)
            return ops.to_nchw(y, 0, C).to(x.dtype)
        if x.dim() == 3:
            B, N, C = x.shape
            return self.forward_tokens(x.to(torch.bfloat16).contiguous().view(B * N, C)).view(B, N, C).to(x.dtype)
        raise ValueError(f"RMSNorm expects 3D or 4D input, got {x.dim()}D")


class TransVAEBlock(nn.Module):
    """x + attn(RMSNorm(x)), then x + ConvFFN(RMSNorm(x))  (R/transvae/modules/blocks.py:89-151).

    ``use_conv_ffn=False`` is shape-broken in the reference and used by no configuration
    (SURVEY.md section 8a-8); it is rejected here.
    """

    def __init__(self, dim: int, mlp_ratio: float = 1.0, head_dim: int = 64, use_rope: bool = True,
                 use_conv_ffn: bool = True, dropout: float = 0.0):
        super().__init__()
        if not use_conv_ffn:
            raise ValueError("TransVAEBlock: only the Conv-FFN variant is implemented (the reference's "
                             "use_conv_ffn=False branch applies nn.Linear to NCHW data and cannot run)")
        if dropout != 0.0:
            raise ValueError("TransVAEBlock: dropout is 0 in every reference configuration and is not implemented")
        self.dim, self.mlp_ratio = dim, mlp_ratio
        self.norm1 = RMSNorm(dim)
        self.attn = FlashAttentionWithRoPE(dim=dim, head_dim=head_dim, use_rope=use_rope, dropout=dropout)
        self.norm2 = RMSNorm(dim)
        self.ffn = ConvFFN(dim=dim, mlp_ratio=mlp_ratio, dropout=dropout)

    def forward_nhwc(self, x: torch.Tensor) -> torch.Tensor:
        B, H, W, C = x.shape
        t = x.view(B * H * W, C)
        t = self.attn.forward_tokens(t, B, H, W, self.norm1.weight, self.norm1.eps)
        t = self.ffn.forward_tokens(t, B, H, W, self.norm2.weight, self.norm2.eps)
        return t.view(B, H, W, C)

    @ops.hip_entry
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return nchw_call(self, x)
