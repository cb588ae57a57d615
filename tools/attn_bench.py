"""Attention kernel timing at the three TransVAE-Large shapes (GPU box).  Diagnostic only."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "deepl-project_amd"))
import torch
from transvae.hip import ops
dev = torch.device("cuda:0")
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 32
tot = 0.0
for (C, N, nblk) in ((384, 4096, 6), (768, 1024, 8), (1536, 256, 12)):
    heads = C // 64
    qkv = torch.randn(mb, N, 3 * C, device=dev).to(torch.bfloat16).requires_grad_(True)
    go = torch.randn(mb, N, C, device=dev).to(torch.bfloat16)
    def fwd():
        return ops.attention(qkv, None, heads, 0.125)
    def fb():
        o = ops.attention(qkv, None, heads, 0.125); o.backward(go); qkv.grad = None
    def tm(fn, it=20):
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(it): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / it
    tf, tfb = tm(fwd), tm(fb)
    flop = 4.0 * mb * heads * N * N * 64
    print(f"C={C:5d} N={N:5d}: fwd {tf:7.3f} ms {flop/tf/1e9:6.0f} TF/s | bwd {tfb-tf:7.3f} ms {2.5*flop/(tfb-tf)/1e9:6.0f} TF/s (2.5x-fwd flops)")
    tot += nblk * tfb
print(f"attention total per micro-batch of {mb}: {tot:.1f} ms")
