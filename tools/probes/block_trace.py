"""Kernel-by-kernel time line of ONE TransVAE block (forward + backward) at a stage's shape, from torch.profiler's device
events: python tools/probes/block_trace.py [dim] [tokens_side] [images] -- e.g. 384 64 64 = stage 2 of Large at 256 px."""
import collections, os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "deepl-project_amd"))
import torch
from transvae.modules.blocks import TransVAEBlock
from transvae.hip import ops
dim = int(sys.argv[1]) if len(sys.argv) > 1 else 384
side = int(sys.argv[2]) if len(sys.argv) > 2 else 64
B = int(sys.argv[3]) if len(sys.argv) > 3 else 64
dev = torch.device("cuda:0")
torch.manual_seed(0)
blk = TransVAEBlock(dim=dim).to(dev)
with torch.no_grad():
    for n, p in blk.named_parameters():
        if p.dim() >= 2:
            p.copy_(torch.randn_like(p) * (p[0].numel()) ** -0.5)
x = torch.randn(B, side, side, dim, device=dev).to(torch.bfloat16).requires_grad_(True)
gy = torch.randn(B, side, side, dim, device=dev).to(torch.bfloat16)
def it():
    y = blk.forward_nhwc(x)
    y.backward(gy)
    x.grad = None
    for p in blk.parameters():
        p.grad = None
for _ in range(3):
    it()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
N = 5
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(N):
        it()
    torch.cuda.synchronize()
ev = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
ev.sort(key=lambda e: e.time_range.start)
per = len(ev) // N
print(f"{len(ev)} device events, {per} per iteration")
tot = 0.0
rows = []
for i in range(per):
    names = {ev[k * per + i].name for k in range(N)}
    t = sum((ev[k * per + i].device_time if hasattr(ev[k * per + i], "device_time") else ev[k * per + i].cuda_time) for k in range(N)) / N
    nm = ev[i].name
    nm = nm.replace("(anonymous namespace)::", "").replace("void ", "")
    rows.append((i, t, nm[:110], len(names)))
    tot += t
for i, t, nm, k in rows:
    if t >= 15:
        print(f"{i:4d} {t:9.1f} us  {nm}{'  (!order varies)' if k > 1 else ''}")
small = sum(t for _, t, _, _ in rows if t < 15)
print(f"kernels under 15 us: {sum(1 for r in rows if r[1] < 15)} launches, {small:.1f} us;  TOTAL {tot / 1e3:.3f} ms per block forward+backward")
