"""Launch census of ONE train step (Large, global batch 4 in micro-batches of 1: one autograd micro-batch + three in-place
ones, FusedAdamW): device kernels by name class, and the aten ops that launch the small torch kernels attributed to the
nearest frame of this repo.  Diagnostic (GPU box):  python tools/probes/launch_census.py [variant]"""
import collections, os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "deepl-project_amd"))
import torch
import bench
from transvae import TransVAE
from transvae.optim import FusedAdamW
from transvae.parallel import train_step, vae_bench_loss
dev = torch.device("cuda:0")
variant = sys.argv[1] if len(sys.argv) > 1 else "large"
with torch.device(dev):
    m = TransVAE(variant=variant, compression_ratio=16, latent_dim=32, clamp_latent=True)
bench.init_scaled_(m, 0)
m.train()
opt = FusedAdamW(m.parameters(), lr=1e-4, betas=(0.9, 0.95), weight_decay=0.0)
x = torch.rand(4, 3, 256, 256, device=dev)
gen = torch.Generator(device=dev).manual_seed(1)
def forward_loss(model, xb):
    eps = torch.randn(xb.shape[0], 32, 16, 16, device=dev, generator=gen)
    recon, mu, logvar = model(xb, eps=eps)
    return vae_bench_loss(recon, xb, mu, logvar)
counters = {}
for _ in range(2):
    train_step(m, opt, x, 1, forward_loss, 1.0, 4, counters)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    train_step(m, opt, x, 1, forward_loss, 1.0, 4, counters)
    torch.cuda.synchronize()
kern = collections.Counter()
ktime = collections.Counter()
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CUDA:
        n = e.name
        cls = "torch:" + n.split("<")[0].replace("void ", "")[:70] if ("at::native" in n or "at_cuda" in n or "rocprim" in n or "hipcub" in n) else ("memcpy/memset" if ("Memcpy" in n or "Memset" in n) else "ours")
        kern[cls] += 1
        ktime[cls] += e.device_time if hasattr(e, "device_time") else e.cuda_time
print("device kernels of one step (4 micro-batches):", sum(kern.values()))
for k, c in kern.most_common(20):
    print(f"  {c:6d}  {ktime[k] / 1e3:9.2f} ms  {k}")
ops_ = collections.Counter()
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CPU and e.name in ("aten::zero_", "aten::fill_", "aten::add_", "aten::add", "aten::copy_", "aten::mul", "aten::mul_", "aten::clone", "aten::contiguous", "aten::sum", "aten::index_select", "aten::cat"):
        st = [s for s in (e.stack or []) if "deepl-project_amd" in s or "parallel.py" in s]
        where = st[0].split("deepl-project_amd/")[-1] if st else ("autograd engine / other" )
        ops_[(e.name, where[:110])] += 1
shp = collections.Counter()
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CPU and e.name in ("aten::zero_", "aten::fill_", "aten::add_", "aten::add", "aten::copy_", "aten::mul", "aten::clone", "aten::cat", "aten::sum"):
        shp[(e.name, str(e.input_shapes)[:90])] += 1
print("aten ops by input shapes:")
for (n, sh), c in shp.most_common(70):
    print(f"  {c:5d} {n:12s} {sh}")
print("aten ops by nearest repo frame:")
for (n, s), c in ops_.most_common(45):
    print(f"  {c:5d} {n:18s} {s}")
