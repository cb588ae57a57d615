"""Convolutional FFN on the HIP path (interface mirror of R/transvae/modules/conv.py:26-105).

    u   = GELU(RMSNorm(t) W_in^T + b_in)            [T, 4d]   GEMM, GELU epilogue (RMS weight folded in W_in)
    c   = GELU(u W_1^T + b_1)                       [T, d]    1x1 conv == GEMM
    c   = GELU(conv3x3(c) + b_2)                    [T, d]    implicit-GEMM conv
    u   = u + c W_3^T + b_3                         [T, 4d]   GEMM, inner residual in the epilogue
    out = t + u W_out^T + b_out                     [T, d]    GEMM, block residual in the epilogue

``conv`` is an nn.Sequential with the reference's indices (0, 2, 4 hold weights; 1, 3 are GELUs) so
that state_dict keys match.  Only conv_type='full' exists on this path (the depthwise variant is
unused ablation code in the reference).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from ..hip import fused, ops


class ConvFFN(nn.Module):
    def __init__(self, dim: int, mlp_ratio: float = 1.0, conv_type: str = "full", dropout: float = 0.0):
        super().__init__()
        if conv_type != "full":
            raise ValueError(f"ConvFFN (HIP path): conv_type={conv_type!r} is not implemented; every config uses 'full'")
        self.dim = dim
        hidden = int(dim * mlp_ratio * 4)
        mid = int(dim * mlp_ratio)
        self.hidden_dim, self.conv_hidden = hidden, mid
        self.proj_in = nn.Linear(dim, hidden)
        self.conv = nn.Sequential(
            nn.Conv2d(hidden, mid, 1), nn.GELU(),
            nn.Conv2d(mid, mid, 3, padding=1), nn.GELU(),
            nn.Conv2d(mid, hidden, 1))
        self.proj_out = nn.Linear(hidden, dim)
        self.dropout = nn.Dropout(dropout)

    def _tail(self, r: torch.Tensor, w_in: torch.Tensor, B: int, H: int, W: int, residual):
        hid, mid = self.hidden_dim, self.conv_hidden
        c0, c2, c4 = self.conv[0], self.conv[2], self.conv[4]
        u = ops.linear(r, w_in, self.proj_in.bias, act="gelu")
        c = ops.linear(u, c0.weight.view(mid, hid), c0.bias, act="gelu")
        c = ops.conv(c.view(B, H, W, mid), c2.weight.permute(0, 2, 3, 1), c2.bias, None, "c3s1", "gelu")
        u = ops.linear(c.view(B * H * W, mid), c4.weight.view(hid, mid), c4.bias, residual=u)
        return ops.linear(u, self.proj_out.weight, self.proj_out.bias, residual=residual)

    def forward_tokens(self, t: torch.Tensor, B: int, H: int, W: int, rms_weight: torch.Tensor, rms_eps: float) -> torch.Tensor:
        """t: [B*H*W, d] bf16 residual stream.  Returns t + ffn(RMSNorm(t))."""
        hid, mid = self.hidden_dim, self.conv_hidden
        c0, c2, c4 = self.conv[0], self.conv[2], self.conv[4]
        w_in = fused.fold([self.proj_in.weight], [rms_weight])[0]      # RMSNorm weight folded into proj_in (one launch)
        return fused.ConvFFNBranchFn.apply(t, w_in, self.proj_in.bias,
                                           c0.weight.view(mid, hid), c0.bias, c2.weight.permute(0, 2, 3, 1), c2.bias,
                                           c4.weight.view(hid, mid), c4.bias, self.proj_out.weight, self.proj_out.bias,
                                           B, H, W, rms_eps)

    @ops.hip_entry
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """Reference-style call on [B, C, H, W] (input already normalised)."""
        B, C, H, W = x.shape
        t = ops.to_nhwc(x, C).view(B * H * W, C)
        y = self._tail(t, self.proj_in.weight, B, H, W, None)
        return ops.to_nchw(y.view(B, H, W, C), 0, C).to(x.dtype)
