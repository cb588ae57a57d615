"""Does a GELU layer prefer two co-resident 128x128 blocks per CU (one block's VALU epilogue under the other's MFMAs)
over one 256-wide block?  Forward with GELU + saved derivative, tile forced through the tuning hook.  GPU box."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "deepl-project_amd"))
from transvae.hip import ops
from transvae.hip import _lib as L
dev = torch.device("cuda:0")
lib = L.load()
def tm(fn, it=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
for (M, K, N) in [(262144, 384, 1536), (262144, 1536, 1536), (65536, 768, 3072), (65536, 3072, 3072), (16384, 1536, 6144), (16384, 6144, 6144)]:
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = torch.randn(N, K, device=dev) * K ** -0.5
    b = torch.randn(N, device=dev) * 0.1
    f = 2.0 * M * K * N
    row = []
    for act, tag in ((L.ACT_NONE, "plain"), (L.ACT_GELU, "gelu+deriv")):
        for cfg in ((0, 0), (128, 128), (256, 128)):
            lib.tv_set_igemm_config(cfg[0], cfg[1], 0, 0)
            t = min(tm(lambda: ops.conv_forward(x, w, b, None, "linear", act, "deriv")) for _ in range(2))
            row.append(f"{tag} {cfg[0]}x{cfg[1]}: {t:.3f} ms ({f/t/1e9:.0f})")
    lib.tv_set_igemm_config(0, 0, 0, 0)
    print(f"M={M} K={K} N={N}: " + " | ".join(row), flush=True)
