"""Uninitialised-read hunt: every torch.empty / empty_like / empty_strided buffer of a floating dtype is filled with NaN
before the kernels get it; a kernel that reads what nobody wrote then produces NaN.  Micro model, two train steps."""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "deepl-project_amd"))
import torch
_e, _el, _es = torch.empty, torch.empty_like, torch.empty_strided
POISON = [True]
def _poison(t):
    if POISON[0] and t.is_cuda and t.is_floating_point() and t.numel():
        t.fill_(float("nan"))
    return t
torch.empty = lambda *a, **k: _poison(_e(*a, **k))
torch.empty_like = lambda *a, **k: _poison(_el(*a, **k))
torch.empty_strided = lambda *a, **k: _poison(_es(*a, **k))
from oracle import filler, transvae_oracle as O
from transvae import TransVAE
from transvae.optim import FusedAdamW
from transvae.parallel import train_step, vae_bench_loss
DEV = "cuda:0"
variant = sys.argv[1] if len(sys.argv) > 1 else "micro"
g = torch.Generator().manual_seed(5)
if variant == "micro":
    m = TransVAE(config=dict(O.MICRO), variant="micro", compression_ratio=16, latent_dim=4, clamp_latent=True)
    m.load_state_dict(filler.fill_state_dict(O.state_dict_schema(O.MICRO, latent_dim=4)))
    x = torch.rand(4, 3, 64, 64, generator=g).to(DEV); eps = torch.randn(4, 4, 4, 4, generator=g).to(DEV); mb = 2
else:
    m = TransVAE(variant=variant, compression_ratio=16, latent_dim=32, clamp_latent=True)
    m.load_state_dict(filler.fill_state_dict(O.state_dict_schema(O.variant_config(variant, 16, 32), 32), gains=filler.LARGE_GAINS if variant == "large" else None))
    x = torch.rand(2, 3, 256, 256, generator=g).to(DEV); eps = torch.randn(2, 32, 16, 16, generator=g).to(DEV); mb = 1
m = m.to(DEV); m.train()
# forward hooks: first module whose output holds a NaN
bad = []
def hook(name):
    def f(mod, inp, out):
        o = out[0] if isinstance(out, (tuple, list)) else out
        if torch.is_tensor(o) and not torch.isfinite(o.float()).all() and not bad:
            bad.append(name)
    return f
for n, mod in m.named_modules():
    mod.register_forward_hook(hook(n))
opt = FusedAdamW(m.parameters(), lr=1e-4, betas=(0.9, 0.95), weight_decay=0.0)
cursor = [0]
def forward_loss(model, xb):
    e = eps[cursor[0]:cursor[0] + xb.shape[0]]; cursor[0] += xb.shape[0]
    recon, mu, logvar = model(xb, eps=e)
    return vae_bench_loss(recon, xb, mu, logvar)
counters = {}
for s in range(2):
    cursor[0] = 0
    loss = float(train_step(m, opt, x, mb, forward_loss, 1.0, x.shape[0], counters))
    torch.cuda.synchronize()
    nan_g = [k for k, p in m.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
    nan_p = [k for k, p in m.named_parameters() if not torch.isfinite(p).all()]
    print(f"step {s}: loss {loss}  grad-norm {float(counters['grad_norm'])}  skipped {float(counters['skipped'])}  first non-finite forward output: {bad[:1]}")
    print(f"   parameters with non-finite gradients: {len(nan_g)} {nan_g[:12]}")
    print(f"   non-finite parameters: {len(nan_p)} {nan_p[:6]}")
