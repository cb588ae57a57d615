"""CPU oracle for the TransVAE forward/backward path.

TEST INFRASTRUCTURE ONLY.  This file is a functional, fp32, plain-PyTorch
restatement of the reference's algorithm for the hot path.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it; the product (``deepl-project_amd/``) never does and fails loudly
when its HIP library is missing.

Parity pin: ``oracle/make_goldens.py`` imports the reference's own model files
from ``/root/reference`` (namespace stub), runs them on
key-seeded weights and commits the outputs under ``tests/golden/``;
``tests/test_oracle_vs_golden.py`` checks every function below against those
vectors to 1e-5.

The model is expressed as pure functions over a ``state_dict`` (same keys and
shapes as the reference's ``TransVAE.state_dict()``), so autograd on the
dictionary's tensors yields the parameter gradients the HIP backward is
checked against.  Tensors are NCHW fp32 at every function boundary, like the
reference.  ``R/`` below = ``/root/reference/transvae-implementation/``.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]

# ---------------------------------------------------------------------------
# variant table (R/transvae/models/transvae.py:107-153; dead code there, but the
# README/test_installation constructor style needs it -- SURVEY F2/F3)
# ---------------------------------------------------------------------------
VARIANTS = {
    "tiny_f16d32": dict(depths=[3, 3, 3, 3, 3], base_dims=[128, 128, 256, 256, 512]),
    "base_f16d32": dict(depths=[3, 3, 3, 3, 3], base_dims=[128, 128, 256, 512, 1024]),
    "large_f16d32": dict(depths=[3, 3, 3, 4, 6], base_dims=[192, 192, 384, 768, 1536]),
    "huge_f16d32": dict(depths=[3, 3, 4, 6, 8], base_dims=[256, 256, 512, 1024, 2048]),
    "giant_f16d32": dict(depths=[3, 3, 4, 8, 10], base_dims=[320, 320, 640, 1280, 2560]),
    "large_f8d16": dict(depths=[3, 3, 6, 8], base_dims=[192, 384, 768, 1536]),
}
MICRO = dict(depths=[1, 1, 1, 1, 1], base_dims=[32, 32, 64, 64, 128], mlp_ratio=1.0, head_dim=64)


def variant_config(variant: str, f: int, d: int) -> dict:
    key = f"{variant}_f{f}d{d}"
    if key not in VARIANTS:
        raise ValueError(f"Unknown variant: {variant} with f{f}d{d}")
    cfg = dict(VARIANTS[key])
    cfg.setdefault("mlp_ratio", 1.0)
    cfg.setdefault("head_dim", 64)
    return cfg


# ---------------------------------------------------------------------------
# primitives
# ---------------------------------------------------------------------------
def gn_silu(x: Tensor, w: Tensor, b: Tensor, groups: int = 32, eps: float = 1e-5) -> Tensor:
    """GroupNorm(32) then SiLU -- R/transvae/modules/blocks.py:33,36,60-65."""
    B, C, H, W = x.shape
    xg = x.reshape(B, groups, -1)
    mean = xg.mean(dim=2, keepdim=True)
    var = xg.var(dim=2, unbiased=False, keepdim=True)
    xh = ((xg - mean) * torch.rsqrt(var + eps)).reshape(B, C, H, W)
    h = xh * w.view(1, C, 1, 1) + b.view(1, C, 1, 1)
    return h * torch.sigmoid(h)


def res_block(x: Tensor, sd: SD, p: str) -> Tensor:
    """ResBlock with identity shortcut -- R/transvae/modules/blocks.py:48-68.

    (in==out everywhere in the shipped configs; a 1x1 shortcut conv is applied
    when the key exists, blocks.py:40-46.)
    """
    h = gn_silu(x, sd[p + "norm1.weight"], sd[p + "norm1.bias"])
    h = F.conv2d(h, sd[p + "conv1.weight"], sd[p + "conv1.bias"], padding=1)
    h = gn_silu(h, sd[p + "norm2.weight"], sd[p + "norm2.bias"])
    h = F.conv2d(h, sd[p + "conv2.weight"], sd[p + "conv2.bias"], padding=1)
    if p + "shortcut.weight" in sd:
        w = sd[p + "shortcut.weight"]
        x = F.conv2d(x, w, sd[p + "shortcut.bias"], padding=w.shape[-1] // 2)
    return h + x


def rms_norm_tokens(t: Tensor, w: Tensor, eps: float = 1e-6) -> Tensor:
    """RMSNorm over the channel axis of token-major [B,N,C] data.

    R/transvae/modules/blocks.py:179-194 does this on NCHW (reduction over
    dim 1); eps sits inside the sqrt.
    """
    ms = (t * t).mean(dim=-1, keepdim=True)
    return t * torch.rsqrt(ms + eps) * w


def ln_hat(t: Tensor, eps: float = 1e-5) -> Tensor:
    """LayerNorm without affine (the x-hat shared by norm_q/k/v)."""
    mean = t.mean(dim=-1, keepdim=True)
    var = t.var(dim=-1, unbiased=False, keepdim=True)
    return (t - mean) * torch.rsqrt(var + eps)


def rope_tables(H: int, W: int, inv_freq: Tensor) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """cos/sin tables of the reference's (non-rotation) 2-D RoPE, [N, hd/2] each.

    R/transvae/modules/attention.py:149-185.  For token n=(y,x) the 64-wide
    angle vector is emb = [y*f, y*f, x*f, x*f] (f = inv_freq, 16 values); pair
    p uses emb[2p] for its first output and emb[2p+1] for the second.
    """
    dev = inv_freq.device
    ys = torch.arange(H, dtype=torch.float32, device=dev)
    xs = torch.arange(W, dtype=torch.float32, device=dev)
    yy = ys.repeat_interleave(W)  # n // W
    xx = xs.repeat(H)  # n % W
    yf = torch.outer(yy, inv_freq.float())
    xf = torch.outer(xx, inv_freq.float())
    emb = torch.cat([yf, yf, xf, xf], dim=-1)  # [N, hd]
    th1 = emb[:, 0::2]
    th2 = emb[:, 1::2]
    return th1.cos(), th1.sin(), th2.cos(), th2.sin()


def rope_apply(t: Tensor, tabs) -> Tensor:
    """t: [B, h, N, hd].  out[2p] = a cos1 - b sin1 ; out[2p+1] = a sin2 + b cos2
    (R/transvae/modules/attention.py:172-197)."""
    c1, s1, c2, s2 = tabs
    a = t[..., 0::2]
    b = t[..., 1::2]
    o1 = a * c1 - b * s1
    o2 = a * s2 + b * c2
    return torch.stack([o1, o2], dim=-1).flatten(-2)


SCORE_BYTES_MAX = 16 << 30   # above this the attention scores are formed 2048 query rows at a time (same arithmetic per row)


def attention_tokens(t: Tensor, H: int, W: int, sd: SD, p: str, head_dim: int = 64,
                     use_rope: bool = True) -> Tensor:
    """FlashAttentionWithRoPE on token-major input t=[B,N,C] (the block's
    RMSNorm output).  R/transvae/modules/attention.py:65-104.

    Three LayerNorms (eps 1e-5, affine with bias) feed three bias-free
    projections; written here in the folded form the HIP path uses: one x-hat,
    gamma folded into the weight columns, W.beta as a bias.
    """
    B, N, C = t.shape
    h = C // head_dim
    xh = ln_hat(t)
    qkv = []
    for nm in ("q", "k", "v"):
        Wm = sd[p + f"to_{nm}.weight"]
        g = sd[p + f"norm_{nm}.weight"]
        be = sd[p + f"norm_{nm}.bias"]
        qkv.append(F.linear(xh, Wm * g[None, :], Wm @ be))
    q, k, v = [u.view(B, N, h, head_dim).transpose(1, 2) for u in qkv]
    if use_rope:
        tabs = rope_tables(H, W, sd[p + "rope.inv_freq"])
        q = rope_apply(q, tabs)
        k = rope_apply(k, tabs)
    scale = head_dim ** -0.5
    if B * h * N * N * 4 <= SCORE_BYTES_MAX:
        s = (q @ k.transpose(-1, -2)) * scale
        o = torch.softmax(s, dim=-1) @ v
    else:   # (N = 65 536 at 1024 x 1024: 100 GB of scores -- query rows are independent, so take them 2048 at a time)
        kt = k.transpose(-1, -2)
        o = torch.cat([torch.softmax((q[:, :, i:i + 2048] @ kt) * scale, dim=-1) @ v for i in range(0, N, 2048)], dim=2)
    o = o.transpose(1, 2).reshape(B, N, C)
    return F.linear(o, sd[p + "proj.weight"], sd[p + "proj.bias"])


def gelu(x: Tensor) -> Tensor:
    """exact (erf) GELU -- R/transvae/modules/conv.py:56,86 use the default."""
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def conv_ffn_tokens(t: Tensor, H: int, W: int, sd: SD, p: str) -> Tensor:
    """ConvFFN (conv_type='full') on token-major t=[B,N,d].
    R/transvae/modules/conv.py:79-105."""
    B, N, d = t.shape
    u = gelu(F.linear(t, sd[p + "proj_in.weight"], sd[p + "proj_in.bias"]))  # [B,N,4d]
    us = u.transpose(1, 2).reshape(B, -1, H, W)
    c = gelu(F.conv2d(us, sd[p + "conv.0.weight"], sd[p + "conv.0.bias"]))
    c = gelu(F.conv2d(c, sd[p + "conv.2.weight"], sd[p + "conv.2.bias"], padding=1))
    c = F.conv2d(c, sd[p + "conv.4.weight"], sd[p + "conv.4.bias"])
    us = us + c
    u2 = us.flatten(2).transpose(1, 2)
    return F.linear(u2, sd[p + "proj_out.weight"], sd[p + "proj_out.bias"])


def transvae_block(x: Tensor, sd: SD, p: str, head_dim: int = 64, use_rope: bool = True) -> Tensor:
    """x = x + attn(RMSNorm1(x)); x = x + ffn(RMSNorm2(x)).
    R/transvae/modules/blocks.py:135-151."""
    B, C, H, W = x.shape
    t = x.flatten(2).transpose(1, 2)  # [B,N,C]
    t = t + attention_tokens(rms_norm_tokens(t, sd[p + "norm1.weight"]), H, W, sd, p + "attn.",
                             head_dim, use_rope)
    t = t + conv_ffn_tokens(rms_norm_tokens(t, sd[p + "norm2.weight"]), H, W, sd, p + "ffn.")
    return t.transpose(1, 2).reshape(B, C, H, W)


def downsample(x: Tensor, sd: SD, p: str) -> Tensor:
    """conv3x3 s1 -> SiLU -> conv3x3 s2, plus pixel_unshuffle(2) -> 1x1 (DC path).
    R/transvae/modules/upsample.py:44-66."""
    h = F.conv2d(x, sd[p + "main_path.0.weight"], sd[p + "main_path.0.bias"], padding=1)
    h = F.silu(h)
    h = F.conv2d(h, sd[p + "main_path.2.weight"], sd[p + "main_path.2.bias"], stride=2, padding=1)
    if p + "dc_conv.weight" in sd:
        B, C, H, W = x.shape
        # pixel_unshuffle: out[b, c*4+dy*2+dx, oy, ox] = x[b, c, 2oy+dy, 2ox+dx]
        xd = x.view(B, C, H // 2, 2, W // 2, 2).permute(0, 1, 3, 5, 2, 4).reshape(B, C * 4, H // 2, W // 2)
        h = h + F.conv2d(xd, sd[p + "dc_conv.weight"], sd[p + "dc_conv.bias"])
    return h


def upsample(x: Tensor, sd: SD, p: str) -> Tensor:
    """nearest x2 -> conv3x3 -> SiLU -> conv3x3, plus 1x1 -> pixel_shuffle(2).
    R/transvae/modules/upsample.py:110-128."""
    B, C, H, W = x.shape
    up = x[:, :, :, None, :, None].expand(B, C, H, 2, W, 2).reshape(B, C, 2 * H, 2 * W)
    h = F.conv2d(up, sd[p + "main_path.1.weight"], sd[p + "main_path.1.bias"], padding=1)
    h = F.silu(h)
    h = F.conv2d(h, sd[p + "main_path.3.weight"], sd[p + "main_path.3.bias"], padding=1)
    if p + "dc_conv.weight" in sd:
        d = F.conv2d(x, sd[p + "dc_conv.weight"], sd[p + "dc_conv.bias"])
        Co = d.shape[1] // 4
        # pixel_shuffle: out[b, c, 2y+dy, 2x+dx] = d[b, c*4+dy*2+dx, y, x]
        d = d.view(B, Co, 2, 2, H, W).permute(0, 1, 4, 2, 5, 3).reshape(B, Co, 2 * H, 2 * W)
        h = h + d
    return h


# ---------------------------------------------------------------------------
# encoder / decoder / model
# ---------------------------------------------------------------------------
def _tap(taps, key: str, h: Tensor) -> Tensor:
    """taps: optional dict that receives every intermediate (precision attribution, tests/precision_report.py); a tap may
    also REPLACE the tensor (taps['@' + key] = callable) so that a stage can be fed a chosen input."""
    if taps is not None:
        taps[key] = h.detach()
        sub = taps.get("@" + key)
        if sub is not None:
            h = sub(h)
    return h


def encoder(x: Tensor, sd: SD, cfg: dict, p: str = "encoder.", use_rope: bool = True, taps=None) -> Tensor:
    """R/transvae/models/encoder.py:101-126 (2 CNN stages, then transformer stages)."""
    depths = cfg["depths"]
    hd = cfg.get("head_dim", 64)
    h = _tap(taps, p + "conv_in", F.conv2d(x, sd[p + "conv_in.weight"], sd[p + "conv_in.bias"], padding=1))
    for i, depth in enumerate(depths):
        for j in range(depth):
            bp = f"{p}stages.{i}.{j}."
            h = _tap(taps, bp[:-1], res_block(h, sd, bp) if i < 2 else transvae_block(h, sd, bp, hd, use_rope))
        if i < len(depths) - 1:
            h = _tap(taps, f"{p}downsamples.{i}", downsample(h, sd, f"{p}downsamples.{i}."))
    return h


def decoder(z: Tensor, sd: SD, cfg: dict, p: str = "decoder.", use_rope: bool = True, taps=None) -> Tensor:
    """R/transvae/models/decoder.py:102-132 (mirror: transformer stages first)."""
    depths = cfg["depths"][::-1]
    hd = cfg.get("head_dim", 64)
    n = len(depths)
    h = _tap(taps, p + "conv_in", F.conv2d(z, sd[p + "conv_in.weight"], sd[p + "conv_in.bias"], padding=1))
    for i, depth in enumerate(depths):
        for j in range(depth):
            bp = f"{p}stages.{i}.{j}."
            h = _tap(taps, bp[:-1], transvae_block(h, sd, bp, hd, use_rope) if i < n - 2 else res_block(h, sd, bp))
        if i < n - 1:
            h = _tap(taps, f"{p}upsamples.{i}", upsample(h, sd, f"{p}upsamples.{i}."))
    h = gn_silu(h, sd[p + "norm_out.weight"], sd[p + "norm_out.bias"])
    return F.conv2d(h, sd[p + "conv_out.weight"], sd[p + "conv_out.bias"], padding=1)


def encode(x: Tensor, sd: SD, cfg: dict, taps=None) -> Tuple[Tensor, Tensor]:
    """R/transvae/models/transvae.py:170-184."""
    h = encoder(x, sd, cfg, taps=taps)
    mu = F.conv2d(h, sd["conv_mu.weight"], sd["conv_mu.bias"], padding=1)
    logvar = F.conv2d(h, sd["conv_logvar.weight"], sd["conv_logvar.bias"], padding=1)
    return mu, logvar


def reparameterize(mu: Tensor, logvar: Tensor, eps: Tensor, clamp: bool = False) -> Tensor:
    """z = mu + eps*exp(logvar/2) with a caller-supplied eps (the reference
    draws it from the global RNG, transvae.py:197-199).  clamp=True is the
    patched copy's variant (P/.../transvae.py:186-196)."""
    if clamp:
        logvar = logvar.clamp(-30.0, 20.0)
    return mu + eps * torch.exp(0.5 * logvar)


def decode(z: Tensor, sd: SD, cfg: dict, taps=None) -> Tensor:
    return decoder(z, sd, cfg, taps=taps)


def forward(x: Tensor, sd: SD, cfg: dict, eps: Tensor, clamp: bool = False, taps=None):
    """R/transvae/models/transvae.py:213-242; clamp=True adds P/...:243-245."""
    mu, logvar = encode(x, sd, cfg, taps)
    if clamp:
        mu = mu.clamp(-50, 50)
        logvar = logvar.clamp(-30, 20)
    z = reparameterize(mu, logvar, eps, clamp)
    return decode(z, sd, cfg, taps), mu, logvar


def bench_loss(recon: Tensor, x: Tensor, mu: Tensor, logvar: Tensor, kl_weight: float = 1e-8,
               clamp_logvar: bool = False) -> Tensor:
    """The closed-form part of the reference loss used for the benchmark:
    L1 + kl_weight * KL  (R/transvae/losses/vae_loss.py:83-84,94-96).

    KL = -0.5 * sum(1 + logvar - mu^2 - exp(logvar)) / (batch * H_lat * W_lat)   (vae_loss.py:94-95: the sum over
    the latent CHANNELS stays, batch and latent pixels are averaged).  clamp_logvar=True clamps logvar to [-30, 20]
    first, as the bf16 trainer does before calling the loss (R/train_2.py:316-318).
    """
    l1 = (recon - x).abs().mean()
    if clamp_logvar:
        logvar = logvar.clamp(-30.0, 20.0)
    kl = -0.5 * torch.sum(1 + logvar - mu.pow(2) - logvar.exp()) / (mu.shape[0] * mu.shape[2] * mu.shape[3])
    return l1 + kl_weight * kl


# ---------------------------------------------------------------------------
# state-dict schema (keys and shapes) -- mirrors the module tree of
# R/transvae/models/{transvae,encoder,decoder}.py; checked against the
# reference-generated tests/golden/state_dict_*.json
# ---------------------------------------------------------------------------
def _res_keys(p, c):
    return {p + "norm1.weight": (c,), p + "norm1.bias": (c,), p + "conv1.weight": (c, c, 3, 3),
            p + "conv1.bias": (c,), p + "norm2.weight": (c,), p + "norm2.bias": (c,),
            p + "conv2.weight": (c, c, 3, 3), p + "conv2.bias": (c,)}


def _tvb_keys(p, c, mlp_ratio, head_dim):
    hid = int(c * mlp_ratio * 4)
    ch = int(c * mlp_ratio)
    k = {p + "norm1.weight": (c,)}
    for nm in "qkv":
        k[p + f"attn.norm_{nm}.weight"] = (c,)
        k[p + f"attn.norm_{nm}.bias"] = (c,)
    for nm in "qkv":
        k[p + f"attn.to_{nm}.weight"] = (c, c)
    k[p + "attn.proj.weight"] = (c, c)
    k[p + "attn.proj.bias"] = (c,)
    k[p + "attn.rope.inv_freq"] = (head_dim // 4,)
    k[p + "norm2.weight"] = (c,)
    k[p + "ffn.proj_in.weight"] = (hid, c)
    k[p + "ffn.proj_in.bias"] = (hid,)
    k[p + "ffn.conv.0.weight"] = (ch, hid, 1, 1)
    k[p + "ffn.conv.0.bias"] = (ch,)
    k[p + "ffn.conv.2.weight"] = (ch, ch, 3, 3)
    k[p + "ffn.conv.2.bias"] = (ch,)
    k[p + "ffn.conv.4.weight"] = (hid, ch, 1, 1)
    k[p + "ffn.conv.4.bias"] = (hid,)
    k[p + "ffn.proj_out.weight"] = (c, hid)
    k[p + "ffn.proj_out.bias"] = (c,)
    return k


def state_dict_schema(cfg: dict, latent_dim: int = 32, input_channels: int = 3) -> Dict[str, tuple]:
    depths, dims = cfg["depths"], cfg["base_dims"]
    mr, hd = cfg.get("mlp_ratio", 1.0), cfg.get("head_dim", 64)
    n = len(depths)
    k: Dict[str, tuple] = {}
    k["encoder.conv_in.weight"] = (dims[0], input_channels, 3, 3)
    k["encoder.conv_in.bias"] = (dims[0],)
    for i in range(n):
        for j in range(depths[i]):
            p = f"encoder.stages.{i}.{j}."
            k.update(_res_keys(p, dims[i]) if i < 2 else _tvb_keys(p, dims[i], mr, hd))
    for i in range(n - 1):
        p = f"encoder.downsamples.{i}."
        k[p + "main_path.0.weight"] = (dims[i], dims[i], 3, 3)
        k[p + "main_path.0.bias"] = (dims[i],)
        k[p + "main_path.2.weight"] = (dims[i + 1], dims[i], 3, 3)
        k[p + "main_path.2.bias"] = (dims[i + 1],)
        k[p + "dc_conv.weight"] = (dims[i + 1], dims[i] * 4, 1, 1)
        k[p + "dc_conv.bias"] = (dims[i + 1],)
    k["conv_mu.weight"] = (latent_dim, dims[-1], 3, 3)
    k["conv_mu.bias"] = (latent_dim,)
    k["conv_logvar.weight"] = (latent_dim, dims[-1], 3, 3)
    k["conv_logvar.bias"] = (latent_dim,)
    rd, rdims = depths[::-1], dims[::-1]
    k["decoder.conv_in.weight"] = (rdims[0], latent_dim, 3, 3)
    k["decoder.conv_in.bias"] = (rdims[0],)
    for i in range(n):
        for j in range(rd[i]):
            p = f"decoder.stages.{i}.{j}."
            k.update(_tvb_keys(p, rdims[i], mr, hd) if i < n - 2 else _res_keys(p, rdims[i]))
    for i in range(n - 1):
        p = f"decoder.upsamples.{i}."
        k[p + "main_path.1.weight"] = (rdims[i + 1], rdims[i], 3, 3)
        k[p + "main_path.1.bias"] = (rdims[i + 1],)
        k[p + "main_path.3.weight"] = (rdims[i + 1], rdims[i + 1], 3, 3)
        k[p + "main_path.3.bias"] = (rdims[i + 1],)
        k[p + "dc_conv.weight"] = (rdims[i + 1] * 4, rdims[i], 1, 1)
        k[p + "dc_conv.bias"] = (rdims[i + 1] * 4,)
    k["decoder.norm_out.weight"] = (rdims[-1],)
    k["decoder.norm_out.bias"] = (rdims[-1],)
    k["decoder.conv_out.weight"] = (input_channels, rdims[-1], 3, 3)
    k["decoder.conv_out.bias"] = (input_channels,)
    return k
