"""Downsample / Upsample with the DC (pixel-(un)shuffle) path, on the HIP path.

Interface mirror of R/transvae/modules/upsample.py:22-128 (``main_path`` Sequential indices and
``dc_conv`` keep the reference's state_dict keys).  Nothing is materialised that the reference
materialises for layout reasons:

  Downsample   dc  = 2x2/stride-2 gather conv             == pixel_unshuffle(2) + 1x1   (:60-61)
               out = conv3x3_s2(SiLU(conv3x3(x))) + dc    residual add in the conv epilogue
  Upsample     dc  = GEMM with pixel-shuffled store       == 1x1 + pixel_shuffle(2)     (:121-123)
               h   = SiLU(conv3x3(nearest2(x)))           polyphase form: four 2x2 convs of x (hip/ops.py)
               out = conv3x3(h) + dc                      residual add in the conv epilogue
"""
from __future__ import annotations

import torch
import torch.nn as nn

from ..hip import fused, ops


def _krsc(conv: nn.Conv2d) -> torch.Tensor:
    return conv.weight.permute(0, 2, 3, 1)


class Downsample(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, use_dc_path: bool = True):
        super().__init__()
        self.in_channels, self.out_channels, self.use_dc_path = in_channels, out_channels, use_dc_path
        self.main_path = nn.Sequential(
            nn.Conv2d(in_channels, in_channels, 3, stride=1, padding=1),
            nn.SiLU(),
            nn.Conv2d(in_channels, out_channels, 3, stride=2, padding=1))
        if use_dc_path:
            self.dc_conv = nn.Conv2d(in_channels * 4, out_channels, 1)

    def forward_nhwc(self, x: torch.Tensor) -> torch.Tensor:
        C, Co = self.in_channels, self.out_channels
        wdc = bdc = None
        if self.use_dc_path:
            # pixel_unshuffle channel order is c*4 + dy*2 + dx  ->  taps (dy,dx), channel c
            wdc = self.dc_conv.weight.view(Co, C, 2, 2).permute(0, 2, 3, 1)
            bdc = self.dc_conv.bias
        return fused.DownsampleFn.apply(x, _krsc(self.main_path[0]), self.main_path[0].bias,
                                        _krsc(self.main_path[2]), self.main_path[2].bias, wdc, bdc)

    @ops.hip_entry
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        from .blocks import nchw_call
        return nchw_call(self, x)


class Upsample(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, use_dc_path: bool = True):
        super().__init__()
        self.in_channels, self.out_channels, self.use_dc_path = in_channels, out_channels, use_dc_path
        self.main_path = nn.Sequential(
            nn.Upsample(scale_factor=2, mode="nearest"),
            nn.Conv2d(in_channels, out_channels, 3, stride=1, padding=1),
            nn.SiLU(),
            nn.Conv2d(out_channels, out_channels, 3, stride=1, padding=1))
        if use_dc_path:
            self.dc_conv = nn.Conv2d(in_channels, out_channels * 4, 1)

    def forward_nhwc(self, x: torch.Tensor) -> torch.Tensor:
        C, Co = self.in_channels, self.out_channels
        wdc = bdc = None
        if self.use_dc_path:
            # pixel_shuffle reads output channel c*4 + dy*2 + dx  ->  GEMM columns ordered (dy,dx,c)
            wdc = self.dc_conv.weight.view(Co, 2, 2, C).permute(1, 2, 0, 3).reshape(4 * Co, 1, 1, C)
            bdc = self.dc_conv.bias.view(Co, 2, 2).permute(1, 2, 0).reshape(4 * Co)
        return fused.UpsampleFn.apply(x, _krsc(self.main_path[1]), self.main_path[1].bias,
                                      _krsc(self.main_path[3]), self.main_path[3].bias, wdc, bdc)

    @ops.hip_entry
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        from .blocks import nchw_call
        return nchw_call(self, x)
