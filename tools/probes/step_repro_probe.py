"""Run-to-run reproducibility of two train steps (micro model, FusedAdamW, 2 micro-batches): which gradients / parameters
differ between identical runs in one process."""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "deepl-project_amd"))
import numpy as np, torch
from oracle import filler, transvae_oracle as O
from transvae import TransVAE
from transvae.optim import FusedAdamW
from transvae.parallel import train_step, vae_bench_loss
DEV = "cuda:0"
g = torch.Generator().manual_seed(5)
x = torch.rand(4, 3, 64, 64, generator=g).to(DEV)
eps = torch.randn(4, 4, 4, 4, generator=g).to(DEV)

def run(micro, kind="hip"):
    m = TransVAE(config=dict(O.MICRO), variant="micro", compression_ratio=16, latent_dim=4, clamp_latent=True)
    m.load_state_dict(filler.fill_state_dict(O.state_dict_schema(O.MICRO, latent_dim=4)))
    m = m.to(DEV); m.train()
    if kind == "hip":
        opt = FusedAdamW(m.parameters(), lr=1e-4, betas=(0.9, 0.95), weight_decay=0.0)
    elif kind == "hip_noshadow":
        opt = FusedAdamW(m.parameters(), lr=1e-4, betas=(0.9, 0.95), weight_decay=0.0, bf16_operands=False)
    else:
        opt = torch.optim.AdamW(m.parameters(), lr=1e-4, betas=(0.9, 0.95), weight_decay=0.0, fused=True)
    cursor = [0]
    def forward_loss(model, xb):
        e = eps[cursor[0]:cursor[0] + xb.shape[0]]; cursor[0] += xb.shape[0]
        recon, mu, logvar = model(xb, eps=e)
        return vae_bench_loss(recon, xb, mu, logvar)
    out = []
    counters = {}
    for s in range(2):
        cursor[0] = 0
        loss = float(train_step(m, opt, x, micro, forward_loss, 1.0, 4, counters))
        torch.cuda.synchronize()
        out.append((loss, float(counters["grad_norm"]), {k: p.grad.detach().double().cpu() for k, p in m.named_parameters()},
                    {k: p.detach().double().cpu() for k, p in m.named_parameters()}))
    return out

kind = sys.argv[1] if len(sys.argv) > 1 else "hip"
runs = [run(2, kind), run(2, kind), run(2, kind), run(4, kind)]
print("optimizer kind", kind)
for i, r in enumerate(runs):
    print("run", i, "losses", [o[0] for o in r], "norms", [o[1] for o in r])
a = runs[0]
for j in (1, 2, 3):
    b = runs[j]
    for s in range(2):
        tot_g = sum(float((a[s][2][k] - b[s][2][k]).norm() ** 2) for k in a[s][2]) ** 0.5 / sum(float(a[s][2][k].norm() ** 2) for k in a[s][2]) ** 0.5
        print(f"run 0 vs run {j}, step {s}: rel-L2 difference of ALL gradients {tot_g:.3e}")
        gd, pdiff = [], []
        for k in a[s][2]:
            n = float(a[s][2][k].norm())
            gd.append((float((a[s][2][k] - b[s][2][k]).norm()) / max(n, 1e-30), k, n))
            d = (a[s][3][k] - b[s][3][k]).abs()
            pdiff.append((int((d > 1e-6).sum()), k, a[s][3][k].numel(), float(d.max())))
        gd.sort(reverse=True); pdiff.sort(reverse=True)
        print(f"run 0 vs run {j}, step {s}: largest gradient rel diffs {[(round(e, 8), k, float('%.3g' % n)) for e, k, n in gd[:6]]}")
        print(f"      parameter elements differing by > 1e-6 after the step: total {sum(p[0] for p in pdiff)}; top {pdiff[:8]}")
