"""Multi-head attention with the reference's 2-D RoPE, on the HIP path.

Interface mirror of R/transvae/modules/attention.py (FlashAttentionWithRoPE, RoPE2D and their
parameter / buffer names).  What runs:

    x-hat  = LayerNorm-hat(RMSNorm(x) * w_rms)                 one fused row kernel
    qkv    = x-hat @ [W_q*g_q ; W_k*g_k ; W_v*g_v]^T + [W_q b_q ; ...]   one MFMA GEMM
             (the three LayerNorms of attention.py:39-41,71-73 share x-hat; their affine parts
              are folded into the projection)
    q, k   = RoPE(q), RoPE(k)                                   in place, table driven
    o      = softmax(q k^T / sqrt(64)) v                        flash attention kernel
    out    = x + o @ W_o^T + b_o                                GEMM with residual epilogue

The weight folding is ordinary differentiable PyTorch on the parameters (O(C^2), independent of
the batch); everything that touches activations is a hand-written kernel.
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch
import torch.nn as nn

from ..hip import fused, ops


class RoPE2D(nn.Module):
    """Table builder for the reference's 2-D rotary embedding (attention.py:107-199).

    For token n = (y, x) the angle vector is [y*f, y*f, x*f, x*f] (f = inv_freq, 16 entries);
    pair p uses angle[2p] for its first output and angle[2p+1] for the second -- NOT a rotation
    (SURVEY F7), reproduced as is.
    """

    def __init__(self, dim: int, max_resolution: int = 4096):
        super().__init__()
        assert dim % 2 == 0, "Dimension must be even for RoPE"
        self.dim, self.max_resolution = dim, max_resolution
        half = dim // 2
        self.register_buffer("inv_freq", 1.0 / (10000 ** (torch.arange(0, half, 2).float() / half)))
        self._tabs: Dict[Tuple[int, int, str], torch.Tensor] = {}

    def table(self, H: int, W: int) -> torch.Tensor:
        """[N, 4, dim/2] fp32: cos(th1), sin(th1), cos(th2), sin(th2) with th1 = angle[0::2], th2 = angle[1::2]."""
        key = (H, W, str(self.inv_freq.device))
        tab = self._tabs.get(key)
        if tab is None:
            f = self.inv_freq.float()
            ys = torch.arange(H, device=f.device, dtype=torch.float32).repeat_interleave(W)
            xs = torch.arange(W, device=f.device, dtype=torch.float32).repeat(H)
            yf, xf = torch.outer(ys, f), torch.outer(xs, f)
            ang = torch.cat([yf, yf, xf, xf], dim=-1)
            t1, t2 = ang[:, 0::2], ang[:, 1::2]
            tab = torch.stack([t1.cos(), t1.sin(), t2.cos(), t2.sin()], dim=1).contiguous()
            self._tabs[key] = tab
        return tab

    def _apply(self, fn, *a, **k):  # tables live on the buffer's device
        self._tabs = {}
        return super()._apply(fn, *a, **k)

    @ops.hip_entry
    def forward(self, x: torch.Tensor, H: int, W: int) -> torch.Tensor:
        """Reference-style call on [B, heads, N, dim] (any float dtype); returns the same shape."""
        B, h, N, d = x.shape
        assert N == H * W and d == self.dim == 64, "RoPE2D (HIP path) supports head_dim 64"
        qkv = torch.zeros((B, N, 3, h, 64), dtype=torch.bfloat16, device=x.device)
        qkv[:, :, 0] = x.permute(0, 2, 1, 3)
        from ..hip import _lib
        import ctypes as C
        _lib.check(_lib.load().tv_rope_qk(C.c_void_p(qkv.data_ptr()), C.c_void_p(self.table(H, W).data_ptr()), B, N, h, 0,
                                          C.c_void_p(torch.cuda.current_stream().cuda_stream)), "tv_rope_qk")
        return qkv[:, :, 0].permute(0, 2, 1, 3).to(x.dtype)


class FlashAttentionWithRoPE(nn.Module):
    def __init__(self, dim: int, head_dim: int = 64, use_rope: bool = True, dropout: float = 0.0):
        super().__init__()
        if head_dim != 64:
            raise ValueError("FlashAttentionWithRoPE (HIP path): head_dim must be 64 (every reference config uses 64)")
        if dim % head_dim != 0:
            raise ValueError("dim must be a multiple of head_dim")
        self.dim, self.head_dim = dim, head_dim
        self.num_heads = dim // head_dim
        self.scale = head_dim ** -0.5
        self.use_rope = use_rope
        self.norm_q = nn.LayerNorm(dim)
        self.norm_k = nn.LayerNorm(dim)
        self.norm_v = nn.LayerNorm(dim)
        self.to_q = nn.Linear(dim, dim, bias=False)
        self.to_k = nn.Linear(dim, dim, bias=False)
        self.to_v = nn.Linear(dim, dim, bias=False)
        self.proj = nn.Linear(dim, dim)
        self.dropout = nn.Dropout(dropout)
        if use_rope:
            self.rope = RoPE2D(head_dim)

    def folded_qkv(self):
        """[3C, C] weight and [3C] bias with the LayerNorm affines folded in (fp32, differentiable)."""
        lins, lns = (self.to_q, self.to_k, self.to_v), (self.norm_q, self.norm_k, self.norm_v)
        if self.to_q.weight.is_cuda:      # one launch per projection, forward and backward (fused.FoldFn)
            return fused.fold([l.weight for l in lins], [n.weight for n in lns], [n.bias for n in lns])
        ws, bs = [], []
        for lin, ln in zip(lins, lns):
            ws.append(lin.weight * ln.weight[None, :])
            bs.append(lin.weight @ ln.bias)
        return torch.cat(ws, 0), torch.cat(bs, 0)

    def forward_tokens(self, t: torch.Tensor, B: int, H: int, W: int, rms_weight: torch.Tensor, rms_eps: float) -> torch.Tensor:
        """t: [B*H*W, C] bf16 residual stream.  Returns t + attn(RMSNorm(t))."""
        w, b = self.folded_qkv()
        tab = self.rope.table(H, W) if self.use_rope else None
        return fused.AttnBranchFn.apply(t, rms_weight, w, b, self.proj.weight, self.proj.bias, tab, B, H * W,
                                        self.num_heads, self.scale, rms_eps, self.norm_q.eps)

    @ops.hip_entry
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """Reference-style call: x [B, C, H, W] (already normalised by the caller) -> attention output."""
        B, C, H, W = x.shape
        t = ops.to_nhwc(x, C).view(B * H * W, C)
        xh = _ln_hat_tokens(t, self.norm_q.eps)  # standalone use: no preceding RMSNorm
        w, b = self.folded_qkv()
        qkv = ops.linear(xh, w, b)
        tab = self.rope.table(H, W) if self.use_rope else None
        o = ops.attention(qkv.view(B, H * W, 3 * C), tab, self.num_heads, self.scale)
        y = ops.linear(o.view(B * H * W, C), self.proj.weight, self.proj.bias)
        return ops.to_nchw(y.view(B, H, W, C), 0, C).to(x.dtype)


def _ln_hat_tokens(t: torch.Tensor, eps_ln: float) -> torch.Tensor:
    """LayerNorm-hat of raw tokens through the fused kernel.

    LN-hat(x) = LN-hat(x * r) exactly when the LN epsilon is scaled by r^2 (r = 1/rms); the
    standalone module call (not used by the model) accepts the eps_ln-vs-eps_ln*r^2 difference,
    which is O(eps) relative.
    """
    ones = torch.ones(t.shape[1], device=t.device)
    return ops.rms_ln_hat(t, ones, 1e-12, eps_ln)
