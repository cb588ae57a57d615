"""Where the bf16 path's error against the fp32 CPU oracle comes from (run on the GPU box).

    python tests/precision_report.py [micro|tiny|large] ... [--json OUT]

For every stage boundary of the encoder and decoder (stem, each block, each down/upsample) two numbers:

  cum    rel-L2 of the HIP path's activation against the oracle's at that point (error accumulated so far)
  local  rel-L2 of that ONE block's output when it is fed the ORACLE's input (rounded to bf16): the error the block
         itself injects, separated from what it inherits

plus the end-to-end recon / mu / logvar errors and the decoder alone fed the oracle's z.  Diagnostic only (imports
oracle/ as the checker); the JSON is what profiles/rNN_precision_attribution.json holds.
"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "deepl-project_amd"))
from oracle import filler  # noqa: E402
from oracle import transvae_oracle as O  # noqa: E402
from transvae import TransVAE  # noqa: E402


def l2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm())


def nchw(t):
    """HIP tap [B,H,W,C] bf16 -> [B,C,H,W] fp32 on the CPU."""
    return t.float().permute(0, 3, 1, 2).cpu()


def nhwc16(t, dev):
    return t.permute(0, 2, 3, 1).contiguous().to(dev).to(torch.bfloat16)


def build(name):
    if name == "micro":
        cfg, L, shape = dict(O.MICRO), 4, (2, 3, 64, 64)
        m = TransVAE(config=cfg, variant="micro", latent_dim=L)
    else:
        cfg, L, shape = O.variant_config(name, 16, 32), 32, (1, 3, 256, 256)
        m = TransVAE(variant=name, latent_dim=L)
    sd = filler.fill_state_dict(O.state_dict_schema(cfg, L), gains=filler.LARGE_GAINS if name == "large" else None)
    m.load_state_dict(sd)
    return m.cuda().eval(), sd, cfg, L, shape


def block_of(m, key):
    mod = m
    for part in key.split("."):
        mod = mod[int(part)] if part.isdigit() else getattr(mod, part)
    return mod


def run(name):
    m, sd, cfg, L, shape = build(name)
    x = filler.rand_input(name + ".x", (4,) + shape[1:])[: shape[0]]
    eps = filler.randn_input(name + ".eps", (4, L, shape[2] // 16, shape[3] // 16))[: shape[0]]
    torch.set_num_threads(min(16, os.cpu_count()))
    ref_taps, hip_taps = {}, {}
    t0 = time.time()
    with torch.no_grad():
        r_ref, mu_ref, lv_ref = O.forward(x, sd, cfg, eps, taps=ref_taps)
    t_cpu = time.time() - t0
    with torch.no_grad():
        h = m.encoder.forward_nhwc(x.cuda(), taps=hip_taps)
        mu, lv = m.encode(x.cuda())
        z = m.reparameterize(mu, lv, eps.cuda())
        m.decoder.forward_nhwc(z, taps=hip_taps)
        r = m.decode(z)
    out = {"model": name, "input": list(shape), "end_to_end": {"recon": l2(r, r_ref), "mu": l2(mu, mu_ref), "logvar": l2(lv, lv_ref)},
           "oracle_fwd_s": round(t_cpu, 2), "stages": []}
    print(f"[{name}] rel-L2  recon {out['end_to_end']['recon']:.4f}  mu {out['end_to_end']['mu']:.4f}  "
          f"logvar {out['end_to_end']['logvar']:.4f}   (oracle fwd {t_cpu:.2f}s)")
    z_ref = O.reparameterize(mu_ref, lv_ref, eps)
    with torch.no_grad():
        d_ref = O.decode(z_ref, sd, cfg)
        d = m.decode(z_ref.cuda())
    out["decoder_alone_from_oracle_z"] = l2(d, d_ref)
    print(f"[{name}] decoder alone, fed the oracle's z: recon rel-L2 {out['decoder_alone_from_oracle_z']:.4f}")
    keys = [k for k in ref_taps if not k.startswith("@")]
    prev = None
    for k in keys:
        cum = l2(nchw(hip_taps[k]), ref_taps[k])
        local = None
        if prev is not None and not k.endswith("conv_in"):       # one block, oracle input (bf16-rounded) -> its own error
            with torch.no_grad():
                y = block_of(m, k).forward_nhwc(nhwc16(ref_taps[prev], "cuda"))
            local = l2(nchw(y), ref_taps[k])
        elif k.endswith("conv_in") and k.startswith("encoder"):
            local = cum
        out["stages"].append({"at": k, "cum": cum, "local": local, "shape": list(ref_taps[k].shape)})
        print(f"  {k:28s} cum {cum:.4f}" + (f"   local {local:.4f}" if local is not None else ""))
        prev = k
    # the decoder's first tap follows z, not the encoder's last tap: recompute its local error from the oracle's z
    for s in out["stages"]:
        if s["at"] == "decoder.conv_in":
            with torch.no_grad():
                taps = {}
                m.decoder.forward_nhwc(z_ref.cuda(), taps=taps)
            s["local"] = l2(nchw(taps["decoder.conv_in"]), ref_taps["decoder.conv_in"])
    return out


if __name__ == "__main__":
    args = sys.argv[1:]
    path = None
    if "--json" in args:
        i = args.index("--json")
        path = args[i + 1]
        args = args[:i] + args[i + 2:]
    res = [run(n) for n in (args or ["micro", "tiny"])]
    if path:
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        with open(path, "w") as f:
            json.dump(res, f, indent=1)
