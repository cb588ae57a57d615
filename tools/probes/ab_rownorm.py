"""Row-norm kernels (RMSNorm / RMSNorm -> LayerNorm-hat rows) at the three transformer stages, one build of the library
(TV_HIP_SO selects it):  python tools/probes/ab_rownorm.py [mb]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "deepl-project_amd"))
from transvae.hip import fused
from transvae.hip import _lib as L
dev = torch.device("cuda:0")
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 64
bf = torch.bfloat16


def tm(fn, it=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it


print("library:", L.SO_PATH)
g = torch.Generator(device=dev).manual_seed(0)
tot = 0.0
for (hw, C, n) in [(64, 384, 6), (32, 768, 8), (16, 1536, 12)]:
    T = mb * hw * hw
    x = torch.randn(T, C, device=dev, generator=g).to(bf)
    dy = torch.randn(T, C, device=dev, generator=g).to(bf)
    dres = torch.randn(T, C, device=dev, generator=g).to(bf)
    w = torch.ones(C, device=dev)
    gb = T * C * 2 / 1e9
    for mode in (0, 1):
        ww = w if mode == 1 else None
        tf = min(tm(lambda: fused.rownorm_fwd(x, ww, mode, 1e-6, 1e-5)) for _ in range(3))
        tb = min(tm(lambda: fused.rownorm_bwd(x, ww, dy, dres, mode, 1e-6, 1e-5)) for _ in range(3))
        tot += 2 * n * (tf + tb) / 2      # (each block runs one mode-0 and one mode-1 norm, forward and backward, in both halves of the model)
        y = fused.rownorm_fwd(x, ww, mode, 1e-6, 1e-5); dx, _ = fused.rownorm_bwd(x, ww, dy, dres, mode, 1e-6, 1e-5)
        print(f"rownorm C={C:5d} T={T:7d} mode {mode}: fwd {tf:6.3f} ms ({2 * gb / tf:5.2f} TB/s)  bwd {tb:6.3f} ms ({4 * gb / tb:5.2f} TB/s)   "
              f"sum {float(y.float().sum()):.5e} {float(dx.float().sum()):.5e}", flush=True)
print(f"row norms per micro-batch of {mb}: {tot:.2f} ms")
