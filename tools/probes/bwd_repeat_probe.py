"""Is the backward pass reproducible for FIXED parameters?  Micro model: gradients of repeated forward+backward passes at the
initial parameters and at the parameters after one optimizer step; and the sensitivity to a 1-ulp change of the parameters."""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "deepl-project_amd"))
import torch
from oracle import filler, transvae_oracle as O
from transvae import TransVAE
from transvae.parallel import train_step, vae_bench_loss
DEV = "cuda:0"
g = torch.Generator().manual_seed(5)
x = torch.rand(4, 3, 64, 64, generator=g).to(DEV); eps = torch.randn(4, 4, 4, 4, generator=g).to(DEV)
m = TransVAE(config=dict(O.MICRO), variant="micro", compression_ratio=16, latent_dim=4, clamp_latent=True)
m.load_state_dict(filler.fill_state_dict(O.state_dict_schema(O.MICRO, latent_dim=4)))
m = m.to(DEV); m.train()

def grads():
    m.zero_grad(set_to_none=True)
    recon, mu, logvar = m(x, eps=eps)
    loss = vae_bench_loss(recon, x, mu, logvar)
    loss.backward()
    torch.cuda.synchronize()
    return float(loss), {k: p.grad.detach().double().cpu() for k, p in m.named_parameters()}, recon.detach().double().cpu()

def rel(a, b):
    num = sum(float((a[k] - b[k]).norm() ** 2) for k in a) ** 0.5
    den = sum(float(b[k].norm() ** 2) for k in b) ** 0.5
    return num / den

def report(tag):
    runs = [grads() for _ in range(4)]
    print(tag, "losses", [r[0] for r in runs])
    for j in (1, 2, 3):
        per = sorted(((float((runs[j][1][k] - runs[0][1][k]).norm()) / max(float(runs[0][1][k].norm()), 1e-30), k, float(runs[0][1][k].norm())) for k in runs[0][1]), reverse=True)
        print(f"   repeat {j} vs 0: all-gradient rel-L2 {rel(runs[j][1], runs[0][1]):.3e}; recon max diff {float((runs[j][2] - runs[0][2]).abs().max()):.3e}; "
              f"top tensors {[(float('%.3g' % e), k, float('%.3g' % n)) for e, k, n in per[:4]]}")
    return runs[0]

r0 = report("initial parameters:")
opt = torch.optim.AdamW(m.parameters(), lr=1e-4, betas=(0.9, 0.95), weight_decay=0.0, fused=True)
torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
opt.step()
torch.cuda.synchronize()
r1 = report("after one AdamW step:")
# sensitivity: nudge every parameter by one fp32 ulp at random
with torch.no_grad():
    gen = torch.Generator(device=DEV).manual_seed(1)
    for p in m.parameters():
        up = torch.nextafter(p, torch.full_like(p, float("inf")))
        mask = torch.rand(p.shape, device=DEV, generator=gen) < 0.05
        p.copy_(torch.where(mask, up, p))
r2 = grads()
print(f"5 % of the parameters moved by one fp32 ulp: loss {r1[0]} -> {r2[0]}, all-gradient rel-L2 change {rel(r2[1], r1[1]):.3e}, recon max diff {float((r2[2] - r1[2]).abs().max()):.3e}")
per = sorted(((float((r2[1][k] - r1[1][k]).norm()) / max(float(r1[1][k].norm()), 1e-30), k, float(r1[1][k].norm())) for k in r1[1]), reverse=True)
print("   top tensors", [(float('%.3g' % e), k, float('%.3g' % n)) for e, k, n in per[:8]])
