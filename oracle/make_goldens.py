"""Mint golden vectors from the reference's own CPU path.

TEST INFRASTRUCTURE ONLY.  Runs in the build container where
``/root/reference`` is mounted; never on the GPU box.  Usage::

    python oracle/make_goldens.py            # writes tests/golden/*.npz, *.json

The reference package cannot be imported normally (its ``__init__`` pulls in
``lpips``), so the model sub-packages are loaded through a
namespace stub exactly as SURVEY.md section 8c describes.  Every fixture holds
outputs only: weights and inputs are regenerated from ``oracle/filler.py``.
"""
from __future__ import annotations

import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import filler  # noqa: E402
from oracle import transvae_oracle as O  # noqa: E402

REF = "/root/reference/transvae-implementation"
OUT = os.path.join(ROOT, "tests", "golden")


def import_reference(patched: bool = False):
    root = REF + ("/transvae-implementation_patched" if patched else "")
    for k in [k for k in sys.modules if k == "transvae" or k.startswith("transvae.")]:
        del sys.modules[k]
    pkg = types.ModuleType("transvae")
    pkg.__path__ = [root + "/transvae"]
    sys.modules["transvae"] = pkg
    from transvae.models.transvae import TransVAE
    from transvae.modules.blocks import ResBlock, TransVAEBlock, RMSNorm
    from transvae.modules.attention import FlashAttentionWithRoPE, RoPE2D
    from transvae.modules.conv import ConvFFN
    from transvae.modules.upsample import Downsample, Upsample
    return dict(TransVAE=TransVAE, ResBlock=ResBlock, TransVAEBlock=TransVAEBlock, RMSNorm=RMSNorm,
                FlashAttentionWithRoPE=FlashAttentionWithRoPE, RoPE2D=RoPE2D, ConvFFN=ConvFFN,
                Downsample=Downsample, Upsample=Upsample)


def load_filled(mod: torch.nn.Module, prefix: str) -> None:
    sd = {k: filler.fill_tensor(prefix + k, v.shape) for k, v in mod.state_dict().items()}
    mod.load_state_dict(sd)


def run_module(mod, prefix, x, gy_tag):
    """forward + backward with a fixed upstream gradient; returns dict of arrays."""
    load_filled(mod, prefix)
    x = x.clone().requires_grad_(True)
    y = mod(x)
    gy = filler.randn_input(gy_tag, y.shape)
    y.backward(gy)
    out = {"y": y.detach().numpy(), "dx": x.grad.numpy()}
    for k, p in mod.named_parameters():
        out["d:" + k] = p.grad.numpy()
    return out


def summarize(t: torch.Tensor, n: int = 64, tag: str = "s") -> dict:
    """mean / std / abs-max + n sampled elements (indices are a function of tag)."""
    flat = t.detach().flatten().double()
    g = torch.Generator().manual_seed(1234)
    idx = torch.randint(0, flat.numel(), (n,), generator=g)
    return {"mean": float(flat.mean()), "std": float(flat.std()), "absmax": float(flat.abs().max()),
            "idx": idx.numpy(), "val": flat[idx].float().numpy()}


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    R = import_reference()

    # ---- per-module goldens ------------------------------------------------
    cases = {}
    cases["resblock"] = run_module(R["ResBlock"](64, 64), "resblock.",
                                   filler.randn_input("resblock.x", (2, 64, 16, 16)), "resblock.gy")
    cases["rmsnorm"] = run_module(R["RMSNorm"](128), "rmsnorm.",
                                  filler.randn_input("rmsnorm.x", (2, 128, 8, 8)), "rmsnorm.gy")
    cases["attn128"] = run_module(R["FlashAttentionWithRoPE"](128, 64), "attn128.",
                                  filler.randn_input("attn128.x", (2, 128, 8, 8)), "attn128.gy")
    cases["attn64"] = run_module(R["FlashAttentionWithRoPE"](64, 64), "attn64.",
                                 filler.randn_input("attn64.x", (1, 64, 16, 12)), "attn64.gy")
    cases["convffn"] = run_module(R["ConvFFN"](128), "convffn.",
                                  filler.randn_input("convffn.x", (2, 128, 8, 8)), "convffn.gy")
    cases["tvblock"] = run_module(R["TransVAEBlock"](128), "tvblock.",
                                  filler.randn_input("tvblock.x", (2, 128, 8, 8)), "tvblock.gy")
    cases["down"] = run_module(R["Downsample"](64, 128), "down.",
                               filler.randn_input("down.x", (2, 64, 16, 16)), "down.gy")
    cases["up"] = run_module(R["Upsample"](128, 64), "up.",
                             filler.randn_input("up.x", (2, 128, 8, 8)), "up.gy")
    for name, d in cases.items():
        np.savez_compressed(os.path.join(OUT, f"mod_{name}.npz"), **d)

    # RoPE2D alone (no parameters): two grids, one non-square
    rope = R["RoPE2D"](64)
    rd = {}
    for (H, W) in [(4, 6), (16, 16)]:
        t = filler.randn_input(f"rope.{H}x{W}", (1, 2, H * W, 64)).requires_grad_(True)
        y = rope(t, H, W)
        gy = filler.randn_input(f"rope.gy.{H}x{W}", y.shape)
        y.backward(gy)
        rd[f"y_{H}x{W}"] = y.detach().numpy()
        rd[f"dx_{H}x{W}"] = t.grad.numpy()
    np.savez_compressed(os.path.join(OUT, "mod_rope.npz"), **rd)

    # ---- micro model end-to-end --------------------------------------------
    cfg = dict(O.MICRO)
    model = R["TransVAE"](config=cfg, variant="micro", compression_ratio=16, latent_dim=4)
    load_filled(model, "")
    x = filler.rand_input("micro.x", (2, 3, 64, 64))
    eps = filler.randn_input("micro.eps", (2, 4, 4, 4))
    z_in = filler.randn_input("micro.z", (2, 4, 4, 4))
    md = {}
    with torch.no_grad():
        mu, logvar = model.encode(x)
        md["mu"], md["logvar"] = mu.numpy(), logvar.numpy()
        md["dec_z"] = model.decode(z_in).numpy()
        md["decoder_direct"] = model.decoder(z_in).numpy()
    # forward with captured eps: patch randn_like for the duration of the call
    orig = torch.randn_like
    torch.randn_like = lambda t, **kw: eps.to(t.dtype)
    try:
        out = model(x, return_dict=True)
    finally:
        torch.randn_like = orig
    recon, mu, logvar, z = out["reconstruction"], out["mu"], out["logvar"], out["z"]
    md["recon"], md["z"] = recon.detach().numpy(), z.detach().numpy()
    loss = O.bench_loss(recon, x, mu, logvar)
    loss.backward()
    md["loss"] = np.array(float(loss))
    gsum = {}
    for k, p in model.named_parameters():
        g = p.grad.flatten()
        gsum[k] = {"l2": float(g.double().norm()), "sum": float(g.double().sum()),
                   "head": g[:8].tolist()}
    with open(os.path.join(OUT, "micro_grads.json"), "w") as f:
        json.dump(gsum, f)
    # a few full gradients (first / last conv, one attention weight, one ffn conv)
    for k in ["encoder.conv_in.weight", "decoder.conv_out.weight", "conv_logvar.weight",
              "encoder.stages.2.0.attn.to_q.weight", "encoder.stages.2.0.attn.norm_k.bias",
              "decoder.stages.0.0.ffn.conv.2.weight", "decoder.stages.3.0.norm1.weight",
              "encoder.downsamples.1.dc_conv.weight", "decoder.upsamples.0.dc_conv.weight",
              "encoder.stages.4.0.norm1.weight"]:
        md["g:" + k] = dict(model.named_parameters())[k].grad.numpy()
    # bf16-autocast run of the same forward (how far the reference's own bf16 is from fp32)
    torch.randn_like = lambda t, **kw: eps.to(t.dtype)
    try:
        with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
            o16 = model(x, return_dict=True)
    finally:
        torch.randn_like = orig
    # ... and of the backward: per-parameter rel-L2 / norm ratio of the reference's bf16-autocast gradients
    g32 = {k: p.grad.clone() for k, p in model.named_parameters()}
    model.zero_grad()
    torch.randn_like = lambda t, **kw: eps.to(t.dtype)
    try:
        with torch.autocast("cpu", dtype=torch.bfloat16):
            ob = model(x, return_dict=True)
        O.bench_loss(ob["reconstruction"].float(), x, ob["mu"].float(), ob["logvar"].float()).backward()
    finally:
        torch.randn_like = orig
    dev16 = {}
    for k, p in model.named_parameters():
        n32 = float(g32[k].double().norm())
        if n32 > 1e-7:
            dev16[k] = {"l2rel": float((p.grad.double() - g32[k].double()).norm() / n32),
                        "norm_ratio": float(p.grad.double().norm() / n32)}
    with open(os.path.join(OUT, "micro_grads_ref_bf16_autocast.json"), "w") as f:
        json.dump(dev16, f)
    print("reference bf16-autocast gradient rel-L2: median %.3f max %.3f" % (
        float(np.median([v["l2rel"] for v in dev16.values()])), max(v["l2rel"] for v in dev16.values())))
    md["recon_bf16"] = o16["reconstruction"].float().numpy()
    md["mu_bf16"] = o16["mu"].float().numpy()
    md["logvar_bf16"] = o16["logvar"].float().numpy()
    np.savez_compressed(os.path.join(OUT, "micro_model.npz"), **md)

    # patched copy: clamps (P/.../transvae.py:186-196,243-245) must be no-ops on these weights
    P = import_reference(patched=True)
    pm = P["TransVAE"](config=cfg, variant="micro", compression_ratio=16, latent_dim=4)
    load_filled(pm, "")
    torch.randn_like = lambda t, **kw: eps.to(t.dtype)
    try:
        with torch.no_grad():
            pr, pmu, plv = pm(x)
    finally:
        torch.randn_like = orig
    np.savez_compressed(os.path.join(OUT, "micro_model_patched.npz"), recon=pr.numpy(), mu=pmu.numpy(),
                        logvar=plv.numpy())

    # ---- BASELINE config 1: tiny f16d32 @256, bs4, fp32 (forward summary) ---
    import yaml
    R = import_reference()
    with open(REF + "/configs/transvae_tiny_f16d32.yaml") as f:
        tcfg = yaml.safe_load(f)["model"]
    tm = R["TransVAE"](config=tcfg, variant="tiny", compression_ratio=16, latent_dim=32)
    load_filled(tm, "")
    xt = filler.rand_input("tiny.x", (4, 3, 256, 256))
    et = filler.randn_input("tiny.eps", (4, 32, 16, 16))
    torch.randn_like = lambda t, **kw: et.to(t.dtype)
    try:
        with torch.no_grad():
            tr, tmu, tlv = tm(xt)
    finally:
        torch.randn_like = orig
    td = {}
    for nm, t in (("recon", tr), ("mu", tmu), ("logvar", tlv)):
        for b in range(4):  # per-image summaries: images are independent through the path
            s = summarize(t[b], 64)
            for kk, vv in s.items():
                td[f"{nm}.{b}.{kk}"] = np.asarray(vv)
    # the reference's own bf16-autocast forward of the same inputs: how far ITS bf16 tier is from fp32
    torch.randn_like = lambda t, **kw: et[:2].to(t.dtype)
    try:
        with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
            ar, amu, alv = tm(xt[:2])
    finally:
        torch.randn_like = orig
    for nm, t16, t32 in (("recon", ar, tr), ("mu", amu, tmu), ("logvar", alv, tlv)):
        for b in range(2):
            td[f"{nm}.{b}.bf16_autocast_l2rel"] = np.asarray(
                float((t16[b].float() - t32[b]).norm() / t32[b].norm()))
    np.savez_compressed(os.path.join(OUT, "tiny_forward.npz"), **td)
    print("tiny reference bf16-autocast vs fp32 rel-L2:",
          {k: float(v) for k, v in td.items() if k.endswith("l2rel")})

    # ---- state-dict schemas from the reference ------------------------------
    schemas = {}
    with open(REF + "/configs/transvae_large_f16d32.yaml") as f:
        lcfg = yaml.safe_load(f)["model"]
    with torch.device("meta"):
        lm = R["TransVAE"](config=lcfg, variant="large", compression_ratio=16, latent_dim=32)
    schemas["large_f16d32"] = {k: list(v.shape) for k, v in lm.state_dict().items()}
    schemas["tiny_f16d32"] = {k: list(v.shape) for k, v in tm.state_dict().items()}
    schemas["micro"] = {k: list(v.shape) for k, v in model.state_dict().items()}
    with open(os.path.join(OUT, "state_dict_schemas.json"), "w") as f:
        json.dump(schemas, f)
    counts = {k: int(sum(int(np.prod(s)) for kk, s in v.items() if not kk.endswith("inv_freq")))
              for k, v in schemas.items()}
    with open(os.path.join(OUT, "param_counts.json"), "w") as f:
        json.dump(counts, f)
    print("param counts", counts)
    print("micro: mu std %.3f logvar std %.3f recon std %.3f loss %.5f" %
          (md["mu"].std(), md["logvar"].std(), md["recon"].std(), float(loss)))
    print("tiny : mu std %.3f logvar std %.3f recon std %.3f" % (tmu.std(), tlv.std(), tr.std()))
    print("micro bf16-autocast rel err: recon %.2e mu %.2e" % (
        np.linalg.norm(md["recon_bf16"] - md["recon"]) / np.linalg.norm(md["recon"]),
        np.linalg.norm(md["mu_bf16"] - md["mu"]) / np.linalg.norm(md["mu"])))


if __name__ == "__main__":
    main()
