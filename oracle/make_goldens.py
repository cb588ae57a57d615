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


LARGE_GRAD_KEYS = [
    "encoder.conv_in.weight", "encoder.stages.0.0.conv1.weight", "encoder.stages.1.2.norm2.weight",
    "encoder.downsamples.0.main_path.2.weight", "encoder.stages.2.0.attn.to_q.weight", "encoder.stages.2.0.attn.norm_k.bias",
    "encoder.stages.2.1.ffn.proj_in.weight", "encoder.stages.4.5.ffn.conv.2.weight", "encoder.downsamples.3.dc_conv.weight",
    "conv_mu.weight", "conv_logvar.weight", "decoder.conv_in.weight", "decoder.stages.0.0.ffn.conv.2.weight",
    "decoder.stages.0.3.attn.proj.weight", "decoder.stages.2.2.attn.to_q.weight", "decoder.upsamples.3.main_path.1.weight",
    "decoder.upsamples.0.dc_conv.weight", "decoder.stages.4.2.conv2.weight", "decoder.norm_out.weight", "decoder.conv_out.weight",
]


def large():
    """BASELINE config 2's model at full size (TransVAE-Large f16d32, 256x256), ONE image, forward + backward through the
    reference on the CPU (fp32), filler weights: output summaries, norms and 256 sampled elements of 20 named gradients
    (N = 4096 attention projections, 1536-wide 3x3 FFN convs, stem, heads, DC paths), and the deviation of the reference's
    OWN bf16-autocast run from its fp32 run on the same tensors (the yardstick of the bf16 tier)."""
    import yaml
    R = import_reference()
    with open(REF + "/configs/transvae_large_f16d32.yaml") as f:
        lcfg = yaml.safe_load(f)["model"]
    model = R["TransVAE"](config=lcfg, variant="large", compression_ratio=16, latent_dim=32)
    load_filled(model, "")
    with torch.no_grad():      # a log-variance head of standard deviation ~1 (oracle/filler.py: LARGE_GAINS)
        for k, g in filler.LARGE_GAINS.items():
            dict(model.named_parameters())[k].mul_(g)
    x = filler.rand_input("large.x", (1, 3, 256, 256))
    eps = filler.randn_input("large.eps", (1, 32, 16, 16))
    orig = torch.randn_like
    out = {}

    def run(autocast):
        model.zero_grad()
        torch.randn_like = lambda t, **kw: eps.to(t.dtype)
        try:
            if autocast:
                with torch.autocast("cpu", dtype=torch.bfloat16):
                    o = model(x, return_dict=True)
            else:
                o = model(x, return_dict=True)
        finally:
            torch.randn_like = orig
        loss = O.bench_loss(o["reconstruction"].float(), x, o["mu"].float(), o["logvar"].float())
        loss.backward()
        params = dict(model.named_parameters())
        return ({k: o[k].detach().float().clone() for k in ("reconstruction", "mu", "logvar")}, float(loss),
                {k: params[k].grad.detach().clone() for k in LARGE_GRAD_KEYS})
    import time
    t0 = time.time()
    o32, loss32, g32 = run(False)
    print("large fp32 fwd+bwd %.1f s, loss %.5f" % (time.time() - t0, loss32), flush=True)
    out["loss"] = np.asarray(loss32)
    for nm, t in (("recon", o32["reconstruction"]), ("mu", o32["mu"]), ("logvar", o32["logvar"])):
        for kk, vv in summarize(t[0], 256).items():
            out[f"{nm}.{kk}"] = np.asarray(vv)
        out[f"{nm}.l2"] = np.asarray(float(t.double().norm()))
    for k, g in g32.items():
        flat = g.flatten()
        gi = torch.Generator().manual_seed(zlib_crc(k))
        idx = torch.randperm(flat.numel(), generator=gi)[:256]
        out[f"g:{k}.idx"] = idx.numpy()
        out[f"g:{k}.val"] = flat[idx].numpy()
        out[f"g:{k}.l2"] = np.asarray(float(flat.double().norm()))
    t0 = time.time()
    o16, loss16, g16 = run(True)
    print("large bf16-autocast fwd+bwd %.1f s" % (time.time() - t0), flush=True)
    dev = {}
    for nm, key in (("recon", "reconstruction"), ("mu", "mu"), ("logvar", "logvar")):
        dev[nm] = float((o16[key].double() - o32[key].double()).norm() / o32[key].double().norm())
    for k in LARGE_GRAD_KEYS:
        n = float(g32[k].double().norm())
        dev["g:" + k] = float((g16[k].double() - g32[k].double()).norm() / n) if n > 1e-12 else 0.0
    print("reference bf16-autocast deviation at Large:", {k: round(v, 4) for k, v in dev.items()})
    np.savez_compressed(os.path.join(OUT, "large_one_image.npz"), **out)
    with open(os.path.join(OUT, "large_ref_bf16_autocast.json"), "w") as f:
        json.dump(dev, f, indent=0)


def extra():
    """The reference's own bf16-autocast deviation (forward outputs + first / last layer gradients) on the two small
    configurations that tests/test_model_gpu.py checks against the oracle without a golden: the micro model at the
    non-square 96 x 160 resolution and the compression-ratio-8 (four-stage) layout.  Inputs are the tests' own seeds."""
    res = {}
    R = import_reference()
    orig = torch.randn_like

    def run(model, x, eps, autocast):
        model.zero_grad()
        torch.randn_like = lambda t, **kw: eps.to(t.dtype)
        try:
            if autocast:
                with torch.autocast("cpu", dtype=torch.bfloat16):
                    o = model(x, return_dict=True)
            else:
                o = model(x, return_dict=True)
        finally:
            torch.randn_like = orig
        O.bench_loss(o["reconstruction"].float(), x, o["mu"].float(), o["logvar"].float()).backward()
        ps = dict(model.named_parameters())
        return ({k: o[k].detach().float().clone() for k in ("reconstruction", "mu", "logvar")},
                {k: ps[k].grad.detach().clone() for k in ("decoder.conv_out.weight", "encoder.conv_in.weight")})

    def dev(model, x, eps):
        o32, g32 = run(model, x, eps, False)
        o16, g16 = run(model, x, eps, True)
        d = {nm: float((o16[k].double() - o32[k].double()).norm() / o32[k].double().norm())
             for nm, k in (("recon", "reconstruction"), ("mu", "mu"), ("logvar", "logvar"))}
        for k in g32:
            d["g:" + k] = float((g16[k].double() - g32[k].double()).norm() / g32[k].double().norm())
        return d
    # micro @ 96 x 160 (test_micro_model_nonsquare_against_oracle: generator seed 7)
    cfg = dict(O.MICRO)
    m = R["TransVAE"](config=cfg, variant="micro", compression_ratio=16, latent_dim=4)
    load_filled(m, "")
    g = torch.Generator().manual_seed(7)
    x = torch.rand(1, 3, 96, 160, generator=g)
    eps = torch.randn(1, 4, 6, 10, generator=g)
    res["micro_96x160"] = dev(m, x, eps)
    # compression ratio 8 (test_f8_style_config_against_oracle: filler keyed on the state-dict keys, generator seed 11)
    cfg8 = dict(depths=[1, 1, 1, 1], base_dims=[32, 64, 64, 128], mlp_ratio=1.0, head_dim=64)
    m8 = R["TransVAE"](config=cfg8, variant="micro8", compression_ratio=8, latent_dim=4)
    load_filled(m8, "")
    g = torch.Generator().manual_seed(11)
    x = torch.rand(2, 3, 64, 64, generator=g)
    eps = torch.randn(2, 4, 8, 8, generator=g)
    res["f8_micro"] = dev(m8, x, eps)
    print(json.dumps(res, indent=1))
    with open(os.path.join(OUT, "ref_bf16_autocast_extra.json"), "w") as f:
        json.dump(res, f, indent=0)


NANO = dict(depths=[1, 1, 1], base_dims=[32, 32, 64], mlp_ratio=1.0, head_dim=64)    # 0.44 M parameters, compression ratio 4


def checkpoint():
    """A checkpoint file written from the REFERENCE model in the reference trainer's format (R/train.py:753-769:
    {'epoch', 'global_step', 'model_state_dict', 'optimizer_state_dict', 'args'}, torch.save) after one real optimizer step
    (fp32 CPU, AdamW lr 1e-3 betas (0.9, 0.95) wd 0.01, clip 1.0, L1 + 1e-8 KL), plus what the reference does NEXT: the loss
    of its second step and samples of the parameters / moments after it -- so a test can resume from the file with the HIP
    model + transvae.optim.FusedAdamW and must land where the reference lands."""
    R = import_reference()
    model = R["TransVAE"](config=dict(NANO), variant="nano", compression_ratio=4, latent_dim=4)
    load_filled(model, "")
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, betas=(0.9, 0.95), weight_decay=0.01)
    orig = torch.randn_like
    losses = []
    for step in range(2):
        x = filler.rand_input(f"nano.x{step}", (2, 3, 32, 32))
        eps = filler.randn_input(f"nano.eps{step}", (2, 4, 8, 8))
        torch.randn_like = lambda t, **kw: eps.to(t.dtype)
        try:
            recon, mu, logvar = model(x)
        finally:
            torch.randn_like = orig
        loss = O.bench_loss(recon, x, mu, logvar)
        opt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        opt.step()
        losses.append(float(loss))
        if step == 0:
            ckpt = {"epoch": 0, "global_step": 1, "model_state_dict": model.state_dict(),
                    "optimizer_state_dict": opt.state_dict(),
                    "args": {"variant": "nano", "compression_ratio": 4, "latent_dim": 4, "learning_rate": 1e-3, "grad_clip": 1.0}}
            torch.save(ckpt, os.path.join(OUT, "ref_checkpoint_nano.pth"))
    expect = {"loss_step0": losses[0], "loss_step1": losses[1], "params": {}, "exp_avg": {}}
    names = [k for k, _ in model.named_parameters()]
    for i, (k, p) in enumerate(model.named_parameters()):
        if i % 7 == 0 or k in ("encoder.conv_in.weight", "decoder.conv_out.weight"):
            flat = p.detach().flatten()
            idx = torch.randperm(flat.numel(), generator=torch.Generator().manual_seed(zlib_crc(k)))[:16]
            expect["params"][k] = {"idx": idx.tolist(), "val": flat[idx].tolist()}
            expect["exp_avg"][k] = {"idx": idx.tolist(), "val": opt.state[p]["exp_avg"].flatten()[idx].tolist()}
    expect["param_order"] = names
    with open(os.path.join(OUT, "ref_checkpoint_nano_expect.json"), "w") as f:
        json.dump(expect, f)
    print("reference checkpoint written; losses", losses)


def _large_model(R, gains=True):
    import yaml
    with open(REF + "/configs/transvae_large_f16d32.yaml") as f:
        lcfg = yaml.safe_load(f)["model"]
    model = R["TransVAE"](config=lcfg, variant="large", compression_ratio=16, latent_dim=32)
    load_filled(model, "")
    if gains:
        with torch.no_grad():      # a log-variance head of standard deviation ~1 (oracle/filler.py: LARGE_GAINS)
            for k, g in filler.LARGE_GAINS.items():
                dict(model.named_parameters())[k].mul_(g)
    return model


def _sample(out, name, t, n=256):
    flat = t.detach().flatten()
    idx = torch.randperm(flat.numel(), generator=torch.Generator().manual_seed(zlib_crc(name)))[:n]
    out[f"{name}.idx"] = idx.numpy()
    out[f"{name}.val"] = flat[idx].float().numpy()
    out[f"{name}.l2"] = np.asarray(float(flat.double().norm()))
    out[f"{name}.std"] = np.asarray(float(flat.double().std()))


LARGE512_GRAD_KEYS = [
    "encoder.conv_in.weight", "encoder.stages.0.1.conv2.weight", "encoder.downsamples.1.main_path.2.weight",
    "encoder.stages.2.0.attn.to_q.weight", "encoder.stages.2.2.attn.norm_k.bias", "encoder.stages.3.1.ffn.proj_in.weight",
    "encoder.stages.4.5.ffn.conv.2.weight", "conv_mu.weight", "conv_logvar.weight", "decoder.conv_in.weight",
    "decoder.stages.0.0.attn.proj.weight", "decoder.stages.2.2.attn.to_k.weight", "decoder.upsamples.2.dc_conv.weight",
    "decoder.stages.4.2.conv2.weight", "decoder.norm_out.weight", "decoder.conv_out.weight",
]


def large512():
    """BASELINE config 4's correctness leg with the backward: TransVAE-Large f16d32 on ONE 512 x 512 image (token grids
    128^2 / 64^2 / 32^2), forward + backward through the REFERENCE on the CPU (fp32, filler weights + LARGE_GAINS): 256
    sampled elements + norms of recon / mu / logvar and of 16 named gradients, and the deviation of the reference's own
    bf16-autocast run on the same tensors.  The reference runs with ITS OWN per-block activation checkpointing
    (R/transvae/models/encoder.py:97-99,117-118) so that the 31 TFLOP fit this container's 64 GB; the values are the same."""
    import time
    R = import_reference()
    model = _large_model(R)
    model.enable_gradient_checkpointing()
    model.train()
    x = filler.rand_input("large512.x", (1, 3, 512, 512))
    eps = filler.randn_input("large512.eps", (1, 32, 32, 32))
    orig = torch.randn_like

    def run(autocast):
        model.zero_grad()
        torch.randn_like = lambda t, **kw: eps.to(t.dtype)
        try:
            if autocast:
                with torch.autocast("cpu", dtype=torch.bfloat16):
                    o = model(x, return_dict=True)
                    loss = O.bench_loss(o["reconstruction"].float(), x, o["mu"].float(), o["logvar"].float())
                    loss.backward()     # (inside the context: the checkpointed blocks re-run under the same autocast state anyway)
            else:
                o = model(x, return_dict=True)
                loss = O.bench_loss(o["reconstruction"].float(), x, o["mu"].float(), o["logvar"].float())
                loss.backward()
        finally:
            torch.randn_like = orig
        params = dict(model.named_parameters())
        return ({k: o[k].detach().float().clone() for k in ("reconstruction", "mu", "logvar")}, float(loss),
                {k: params[k].grad.detach().clone() for k in LARGE512_GRAD_KEYS})
    t0 = time.time()
    o32, loss32, g32 = run(False)
    print("large512 fp32 fwd+bwd %.1f s, loss %.5f" % (time.time() - t0, loss32), flush=True)
    out = {"loss": np.asarray(loss32)}
    for nm, key in (("recon", "reconstruction"), ("mu", "mu"), ("logvar", "logvar")):
        _sample(out, nm, o32[key])
    for k, g in g32.items():
        _sample(out, "g:" + k, g)
    np.savez_compressed(os.path.join(OUT, "large512_one_image.npz"), **out)
    t0 = time.time()
    o16, loss16, g16 = run(True)
    print("large512 bf16-autocast fwd+bwd %.1f s" % (time.time() - t0), flush=True)
    dev = {"loss_bf16": loss16}
    for nm, key in (("recon", "reconstruction"), ("mu", "mu"), ("logvar", "logvar")):
        dev[nm] = float((o16[key].double() - o32[key].double()).norm() / o32[key].double().norm())
    for k in LARGE512_GRAD_KEYS:
        n = float(g32[k].double().norm())
        dev["g:" + k] = float((g16[k].double() - g32[k].double()).norm() / n) if n > 1e-12 else 0.0
    print("reference bf16-autocast deviation at Large 512x512:", {k: round(v, 4) for k, v in dev.items()})
    with open(os.path.join(OUT, "large512_ref_bf16_autocast.json"), "w") as f:
        json.dump(dev, f, indent=0)


def large1024():
    """SURVEY 8f-3 widening (R/scripts/reproduce/test_rope_extrapolation.py:28-51 evaluates 256 / 512 / 1024): TransVAE-Large
    f16d32 on ONE 1024 x 1024 image, forward only (no-grad inference path: encode -> z = mu + eps exp(logvar / 2) -> decode),
    through the reference on the CPU: sampled values + norms, and the reference's own bf16-autocast deviation."""
    import time
    R = import_reference()
    model = _large_model(R)
    model.eval()
    x = filler.rand_input("large1024.x", (1, 3, 1024, 1024))
    eps = filler.randn_input("large1024.eps", (1, 32, 64, 64))
    orig = torch.randn_like

    def run(autocast):
        torch.randn_like = lambda t, **kw: eps.to(t.dtype)
        try:
            with torch.no_grad():
                if autocast:
                    with torch.autocast("cpu", dtype=torch.bfloat16):
                        o = model(x, return_dict=True)
                else:
                    o = model(x, return_dict=True)
        finally:
            torch.randn_like = orig
        return {k: o[k].detach().float().clone() for k in ("reconstruction", "mu", "logvar")}
    t0 = time.time()
    o32 = run(False)
    print("large1024 fp32 forward %.1f s" % (time.time() - t0), flush=True)
    out = {}
    for nm, key in (("recon", "reconstruction"), ("mu", "mu"), ("logvar", "logvar")):
        _sample(out, nm, o32[key], 1024)
    np.savez_compressed(os.path.join(OUT, "large1024_one_image.npz"), **out)
    t0 = time.time()
    o16 = run(True)
    print("large1024 bf16-autocast forward %.1f s" % (time.time() - t0), flush=True)
    dev = {nm: float((o16[key].double() - o32[key].double()).norm() / o32[key].double().norm())
           for nm, key in (("recon", "reconstruction"), ("mu", "mu"), ("logvar", "logvar"))}
    print("reference bf16-autocast deviation at Large 1024x1024:", dev)
    with open(os.path.join(OUT, "large1024_ref_bf16_autocast.json"), "w") as f:
        json.dump(dev, f, indent=0)


def large_unit():
    """The UNIT-GAIN Large fixture (no LARGE_GAINS: logvar of standard deviation ~5, |logvar| up to 20), 256 x 256, ONE image,
    forward through the reference: sampled mu / logvar / z / recon, the decoder's output on the fp32 z, and the reference's
    own bf16-autocast deviation split the way the tests assert it -- encoder outputs, and the DECODER ALONE on a fixed z
    (decode(z32) under autocast vs fp32), i.e. without the exp(logvar / 2) amplification of the encoder's error."""
    import time
    R = import_reference()
    model = _large_model(R, gains=False)
    model.eval()
    x = filler.rand_input("large.x", (1, 3, 256, 256))
    eps = filler.randn_input("large.eps", (1, 32, 16, 16))
    orig = torch.randn_like
    torch.randn_like = lambda t, **kw: eps.to(t.dtype)
    try:
        t0 = time.time()
        with torch.no_grad():
            o32 = model(x, return_dict=True)
            with torch.autocast("cpu", dtype=torch.bfloat16):
                o16 = model(x, return_dict=True)
                d16 = model.decode(o32["z"])
        print("large unit-gain forward x2 + decode %.1f s" % (time.time() - t0), flush=True)
    finally:
        torch.randn_like = orig
    out = {}
    for nm, key in (("recon", "reconstruction"), ("mu", "mu"), ("logvar", "logvar"), ("z", "z")):
        _sample(out, nm, o32[key], 1024)
    out["logvar.absmax"] = np.asarray(float(o32["logvar"].abs().max()))
    np.savez_compressed(os.path.join(OUT, "large_unit_one_image.npz"), **out)

    def rel(a, b):
        return float((a.double() - b.double()).norm() / b.double().norm())
    dev = {"recon": rel(o16["reconstruction"].float(), o32["reconstruction"]), "mu": rel(o16["mu"].float(), o32["mu"]),
           "logvar": rel(o16["logvar"].float(), o32["logvar"]), "z": rel(o16["z"].float(), o32["z"]),
           "decoder_alone": rel(d16.float(), o32["reconstruction"])}
    print("reference bf16-autocast deviation, unit-gain Large:", dev)
    with open(os.path.join(OUT, "large_unit_ref_bf16_autocast.json"), "w") as f:
        json.dump(dev, f, indent=0)


def schemas_more():
    """State-dict schemas / parameter counts of the variants that have no YAML (huge, giant = BASELINE's "XL", large_f8d16),
    built by the reference on the meta device from ITS OWN variant table (R/transvae/models/transvae.py:107-153), merged
    into tests/golden/state_dict_schemas.json and param_counts.json."""
    R = import_reference()
    T = R["TransVAE"]
    with open(os.path.join(OUT, "state_dict_schemas.json")) as f:
        schemas = json.load(f)
    for variant, f_, d in (("huge", 16, 32), ("giant", 16, 32), ("large", 8, 16)):
        cfg = T._get_variant_config(None, variant, f_, d)
        with torch.device("meta"):
            m = T(config=cfg, variant=variant, compression_ratio=f_, latent_dim=d)
        schemas[f"{variant}_f{f_}d{d}"] = {k: list(v.shape) for k, v in m.state_dict().items()}
    with open(os.path.join(OUT, "state_dict_schemas.json"), "w") as f:
        json.dump(schemas, f)
    counts = {k: int(sum(int(np.prod(s)) for kk, s in v.items() if not kk.endswith("inv_freq"))) for k, v in schemas.items()}
    with open(os.path.join(OUT, "param_counts.json"), "w") as f:
        json.dump(counts, f)
    print("param counts", counts)


LARGE_STEP_KEYS = [
    "encoder.conv_in.weight", "encoder.stages.0.0.conv1.weight", "encoder.stages.1.2.norm2.weight",
    "encoder.downsamples.0.main_path.2.weight", "encoder.stages.2.0.attn.to_q.weight", "encoder.stages.2.0.attn.norm_k.bias",
    "encoder.stages.3.1.ffn.proj_in.weight", "encoder.stages.4.5.ffn.conv.2.weight", "conv_logvar.weight",
    "decoder.conv_in.weight", "decoder.stages.0.3.attn.proj.weight", "decoder.upsamples.3.main_path.1.weight",
    "decoder.stages.4.2.conv2.bias", "decoder.conv_out.weight",
]


def large_steps():
    """BASELINE config 2's UNIT OF WORK pinned to the reference: TWO optimizer steps of TransVAE-Large f16d32 on ONE 256 x 256
    image per step, through the reference's patched model (P/.../transvae.py:186-196,243-245: the clamps the bench runs
    with) and the reference trainer's step (R/train.py:577-620,681-687: loss -> backward -> clip_grad_norm_(1.0) ->
    torch.optim.AdamW(lr 1e-4, betas (0.9, 0.95), weight_decay 0); loss = L1 + 1e-8 KL with the logvar clamp of
    R/train_2.py:316-318), fp32 on the CPU, filler weights + LARGE_GAINS.  Recorded per step: loss, gradient norm before the
    clip, and for 14 named tensors 256 sampled gradient values and the parameter delta (p_after - p_before) at the same
    indices.  Then the same two steps under torch.autocast(cpu, bf16) from the same start: the reference's OWN bf16
    deviation of every recorded quantity (the yardstick of the bf16 tier)."""
    import time
    P = import_reference(patched=True)

    def two_steps(autocast):
        model = _large_model(P)
        model.train()
        params = dict(model.named_parameters())
        opt = torch.optim.AdamW(model.parameters(), lr=1e-4, betas=(0.9, 0.95), weight_decay=0.0)
        idx = {k: torch.randperm(params[k].numel(), generator=torch.Generator().manual_seed(zlib_crc("steps:" + k)))[:256]
               for k in LARGE_STEP_KEYS}
        rec = {}
        orig = torch.randn_like
        for step in range(2):
            x = filler.rand_input(f"largesteps.x{step}", (1, 3, 256, 256))
            eps = filler.randn_input(f"largesteps.eps{step}", (1, 32, 16, 16))
            before = {k: params[k].detach().flatten()[idx[k]].clone() for k in LARGE_STEP_KEYS}
            opt.zero_grad(set_to_none=True)
            torch.randn_like = lambda t, **kw: eps.to(t.dtype)
            t0 = time.time()
            try:
                if autocast:
                    with torch.autocast("cpu", dtype=torch.bfloat16):
                        recon, mu, logvar = model(x)
                else:
                    recon, mu, logvar = model(x)
            finally:
                torch.randn_like = orig
            loss = O.bench_loss(recon.float(), x, mu.float(), logvar.float(), clamp_logvar=True)
            loss.backward()
            norm = torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
            grads = {k: params[k].grad.detach().flatten()[idx[k]].clone() for k in LARGE_STEP_KEYS}
            gl2 = {k: float(params[k].grad.double().norm()) for k in LARGE_STEP_KEYS}      # (after the clip, as the optimizer sees them)
            opt.step()
            print("large-steps %s step %d: loss %.6f grad-norm %.4f  %.1f s" % ("bf16" if autocast else "fp32", step, float(loss),
                                                                            float(norm), time.time() - t0), flush=True)
            rec[f"loss{step}"] = float(loss)
            rec[f"gnorm{step}"] = float(norm)
            for k in LARGE_STEP_KEYS:
                rec[f"s{step}.g:{k}"] = grads[k]
                rec[f"s{step}.gl2:{k}"] = gl2[k]
                rec[f"s{step}.d:{k}"] = params[k].detach().flatten()[idx[k]] - before[k]
        return rec, idx
    r32, idx = two_steps(False)
    out = {}
    for k, v in r32.items():
        out[k] = v.numpy() if torch.is_tensor(v) else np.asarray(v)
    for k, i in idx.items():
        out[f"idx:{k}"] = i.numpy()
    np.savez_compressed(os.path.join(OUT, "large_two_steps.npz"), **out)
    r16, _ = two_steps(True)
    dev = {}
    for k, v in r32.items():
        if torch.is_tensor(v):
            n = float(v.double().norm())
            dev[k] = float((r16[k].double() - v.double()).norm() / n) if n > 0 else 0.0
        else:
            dev[k] = abs(r16[k] - v) / abs(v)
    dev["loss_bf16"] = [r16["loss0"], r16["loss1"]]
    dev["gnorm_bf16"] = [r16["gnorm0"], r16["gnorm1"]]
    print("reference bf16-autocast deviation over two Large steps:", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in dev.items()})
    with open(os.path.join(OUT, "large_two_steps_ref_bf16_autocast.json"), "w") as f:
        json.dump(dev, f, indent=0)


TINY_GRAD_KEYS = [
    "encoder.conv_in.weight", "encoder.conv_in.bias", "encoder.stages.0.0.norm1.weight", "encoder.stages.1.0.conv2.weight",
    "encoder.downsamples.1.dc_conv.weight", "encoder.stages.2.0.attn.to_k.weight", "encoder.stages.3.0.ffn.conv.2.weight",
    "encoder.stages.4.0.attn.norm_q.weight", "conv_mu.weight", "conv_logvar.bias", "decoder.conv_in.weight",
    "decoder.stages.0.0.ffn.proj_out.weight", "decoder.stages.2.0.norm2.weight", "decoder.upsamples.1.main_path.3.weight",
    "decoder.stages.4.0.conv1.weight", "decoder.conv_out.weight",
]


def tiny_bs4():
    """BASELINE config 1 at its STATED batch with the backward: tiny f16d32 (R/configs/transvae_tiny_f16d32.yaml), 256 x 256,
    batch 4, fp32, forward + backward of L1 + 1e-8 KL through the reference on the CPU (inputs / noise / weights = the ones of
    tiny_forward.npz): loss, 256 samples + norm of recon / mu / logvar over the whole batch, 256 samples + norm of 16 named
    gradients; and the reference's own bf16-autocast deviation of each."""
    import time
    import yaml
    R = import_reference()
    with open(REF + "/configs/transvae_tiny_f16d32.yaml") as f:
        tcfg = yaml.safe_load(f)["model"]
    tm = R["TransVAE"](config=tcfg, variant="tiny", compression_ratio=16, latent_dim=32)
    load_filled(tm, "")
    tm.train()
    xt = filler.rand_input("tiny.x", (4, 3, 256, 256))
    et = filler.randn_input("tiny.eps", (4, 32, 16, 16))
    orig = torch.randn_like

    def run(autocast):
        tm.zero_grad()
        torch.randn_like = lambda t, **kw: et.to(t.dtype)
        try:
            if autocast:
                with torch.autocast("cpu", dtype=torch.bfloat16):
                    o = tm(xt, return_dict=True)
            else:
                o = tm(xt, return_dict=True)
        finally:
            torch.randn_like = orig
        loss = O.bench_loss(o["reconstruction"].float(), xt, o["mu"].float(), o["logvar"].float())
        loss.backward()
        ps = dict(tm.named_parameters())
        return ({k: o[k].detach().float().clone() for k in ("reconstruction", "mu", "logvar")}, float(loss),
                {k: ps[k].grad.detach().clone() for k in TINY_GRAD_KEYS})
    t0 = time.time()
    o32, loss32, g32 = run(False)
    print("tiny bs4 fp32 fwd+bwd %.1f s, loss %.6f" % (time.time() - t0, loss32), flush=True)
    out = {"loss": np.asarray(loss32)}
    for nm, key in (("recon", "reconstruction"), ("mu", "mu"), ("logvar", "logvar")):
        _sample(out, nm, o32[key])
    for k, g in g32.items():
        _sample(out, "g:" + k, g)
    np.savez_compressed(os.path.join(OUT, "tiny_bs4_fwd_bwd.npz"), **out)
    t0 = time.time()
    o16, loss16, g16 = run(True)
    print("tiny bs4 bf16-autocast fwd+bwd %.1f s" % (time.time() - t0), flush=True)
    dev = {"loss_bf16": loss16}
    for nm, key in (("recon", "reconstruction"), ("mu", "mu"), ("logvar", "logvar")):
        dev[nm] = float((o16[key].double() - o32[key].double()).norm() / o32[key].double().norm())
    for k in TINY_GRAD_KEYS:
        n = float(g32[k].double().norm())
        dev["g:" + k] = float((g16[k].double() - g32[k].double()).norm() / n) if n > 1e-12 else 0.0
    print("reference bf16-autocast deviation, tiny bs4:", {k: round(v, 4) for k, v in dev.items()})
    with open(os.path.join(OUT, "tiny_bs4_ref_bf16_autocast.json"), "w") as f:
        json.dump(dev, f, indent=0)


def zlib_crc(k: str) -> int:
    import zlib
    return zlib.crc32(k.encode()) & 0x7FFFFFFF


if __name__ == "__main__":
    if "--large" in sys.argv:      # 1.05 B parameters through the reference on the CPU: minutes, ~30 GB
        large()
    elif "--extra" in sys.argv:
        extra()
    elif "--checkpoint" in sys.argv:
        checkpoint()
    elif "--large512" in sys.argv:   # ONE 512 x 512 image, forward + backward: ~10 min, < 40 GB (reference checkpointing on)
        large512()
    elif "--large1024" in sys.argv:  # ONE 1024 x 1024 image, forward only: ~20 min
        large1024()
    elif "--large-unit" in sys.argv:
        large_unit()
    elif "--schemas-more" in sys.argv:
        schemas_more()
    elif "--large-steps" in sys.argv:  # two reference optimizer steps of Large on one 256 x 256 image, fp32 + bf16: ~10 min, ~25 GB
        large_steps()
    elif "--tiny-bs4" in sys.argv:     # config 1 at batch 4, forward + backward, fp32 + bf16: ~3 min
        tiny_bs4()
    else:
        main()
