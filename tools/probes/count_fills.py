"""Where do the small fill / add kernels of a micro-batch come from?  (diagnostic; GPU box)"""
import collections, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "deepl-project_amd"))
from transvae import TransVAE
from transvae.parallel import vae_bench_loss
dev = torch.device("cuda:0")
m = TransVAE(variant="tiny", compression_ratio=16, latent_dim=32).to(dev)
x = torch.rand(4, 3, 256, 256, device=dev)
def step():
    recon, mu, logvar = m(x)
    loss = vae_bench_loss(recon, x, mu, logvar)
    loss.backward()
step(); step()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    step()
cnt = collections.Counter()
for e in prof.events():
    if e.name in ("aten::zero_", "aten::fill_", "aten::zeros", "aten::add_", "aten::add"):
        st = [s for s in (e.stack or []) if "deepl-project_amd" in s or "parallel.py" in s or "autograd" in s]
        cnt[(e.name, st[0] if st else "(no repo frame)")] += 1
for (n, s), c in cnt.most_common(25):
    print(f"{c:5d} {n:12s} {s[:150]}")
