"""Cost of the activation epilogue: the same linear layer with and without GELU (+ saved pre-activation).  GPU box."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "deepl-project_amd"))
from transvae.hip import ops
from transvae.hip import _lib as L
dev = torch.device("cuda:0")
def tm(fn, it=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
for (M, K, N) in [(65536, 768, 3072), (16384, 1536, 6144), (262144, 384, 1536), (65536, 3072, 768)]:
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = torch.randn(N, K, device=dev) * K ** -0.5
    b = torch.randn(N, device=dev) * 0.1
    res = torch.randn(M, N, device=dev).to(torch.bfloat16)
    f = 2.0 * M * K * N
    t0 = tm(lambda: ops.conv_forward(x, w, b, None, "linear", L.ACT_NONE, False))
    t1 = tm(lambda: ops.conv_forward(x, w, b, None, "linear", L.ACT_GELU, False))
    t2 = tm(lambda: ops.conv_forward(x, w, b, None, "linear", L.ACT_GELU, True))
    t3 = tm(lambda: ops.conv_forward(x, w, b, res, "linear", L.ACT_NONE, False))
    print(f"M={M} K={K} N={N}: plain {t0:.3f} ms ({f/t0/1e9:.0f} TF/s) | +GELU {t1:.3f} | +GELU+pre {t2:.3f} ({f/t2/1e9:.0f} TF/s) | +residual {t3:.3f}")
