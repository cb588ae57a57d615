"""Fixed costs of one igemm tile on the K = 384 -> N = 1536 linear layer at 64 images, timed with HIP events around 20
back-to-back C-ABI launches (no Python op in the timed region).  TV_HIP_SO selects the build."""
import os, sys, ctypes as C, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "deepl-project_amd"))
from transvae.hip import ops
from transvae.hip import _lib as L
dev = torch.device("cuda:0")
bf = torch.bfloat16
print("library:", L.SO_PATH)
for (hw, Cin, Cout) in [(64, 384, 1536), (32, 768, 3072), (16, 1536, 6144), (64, 1536, 384)]:
    M = 64 * hw * hw
    x = torch.randn(M, Cin, device=dev).to(bf)
    w = torch.randn(Cout, Cin, device=dev) * Cin ** -0.5
    b = torch.randn(Cout, device=dev) * 0.1
    for name, act, want in (("plain", L.ACT_NONE, False), ("gelu+deriv", L.ACT_GELU, "deriv")):
        fn = lambda: ops.conv_forward(x, w, b, None, "linear", act, want)[0]
        for _ in range(3): fn()
        torch.cuda.synchronize()
        with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
            for _ in range(10): fn()
            torch.cuda.synchronize()
        ts = [e.device_time_total / max(e.count, 1) for e in prof.key_averages() if "igemm" in e.key or "halo" in e.key]
        print(f"{Cin}->{Cout}@{hw} {name:11s} kernel {sum(ts):8.1f} us", flush=True)
