"""GroupNorm(32) + SiLU passes alone at the two ResBlock shapes (64 images): ms and TB/s per kernel, for A/B builds of
norm.hip (TV_HIP_SO=tools/probes/abl/lib_X.so python tools/probes/gn_bench.py)."""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "deepl-project_amd"))
import torch
from transvae.hip import fused, _lib as L
dev = torch.device("cuda:0")
print("library:", L.SO_PATH)
def tm(fn, it=20):
    for _ in range(4): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
tot = 0.0
for res in (256, 128):
    B, C = 64, 192
    g = torch.Generator(device=dev).manual_seed(res)
    x = torch.randn(B, res, res, C, device=dev, generator=g).to(torch.bfloat16)
    dy = torch.randn(B, res, res, C, device=dev, generator=g).to(torch.bfloat16)
    dres = torch.randn(B, res, res, C, device=dev, generator=g).to(torch.bfloat16)
    ga = 1 + 0.1 * torch.randn(C, device=dev, generator=g); be = 0.1 * torch.randn(C, device=dev, generator=g)
    gb = x.numel() * 2 / 1e9
    y, mr = fused.gn_silu_fwd(x, ga, be, 32, 1e-5)
    rows = [("fwd (stats + apply: 3 passes)", lambda: fused.gn_silu_fwd(x, ga, be, 32, 1e-5), 3 * gb),
            ("apply from mean/rstd (2 passes)", lambda: fused.gn_silu_apply(x, mr, ga, be, 32), 2 * gb),
            ("bwd reduce + apply (5 passes)", lambda: fused.gn_silu_bwd(x, dy, None, mr, ga, be, 32), 5 * gb),
            ("bwd reduce + apply + residual (6 passes)", lambda: fused.gn_silu_bwd(x, dy, dres, mr, ga, be, 32), 6 * gb)]
    for name, fn, byt in rows:
        t = min(tm(fn) for _ in range(3))
        tot += t
        print(f"192@{res:<3d} {name:42s} {t:7.3f} ms  {byt / t:6.2f} TB/s")
    del x, dy, dres, y
print(f"SUM {tot:.3f} ms")
