"""Fixed per-tile cost of igemm_nt: same M, N as the dominant conv but K = 64 .. 1728 (linear layers)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "deepl-project_amd"))
import torch
from transvae.hip import ops
dev = torch.device("cuda:0")
M, N = 32 * 256 * 256, 192
for K in (64, 128, 256, 576, 1728):
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = torch.randn(N, 1, K, device=dev)
    wb, _ = ops.pack_weight(w, True, False, False)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    d = ops._desc(batch=M, h_in=1, w_in=1, c_in=K, ldx=K, h_out=1, w_out=1, c_out=N, ldo=N, kh=1, kw=1)
    for _ in range(2): ops.igemm(d, x, wb, None, None, None, out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): ops.igemm(d, x, wb, None, None, None, out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"K={K:5d}  {ms:7.3f} ms   {2.0*M*N*K/ms/1e9:6.0f} TF/s   HBM {(M*K*2+M*N*2)/ms/1e6:6.0f} GB/s")
