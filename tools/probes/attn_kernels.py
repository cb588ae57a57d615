"""Per-kernel times of the attention kernels (forward, dq, dk/dv) at the three TransVAE-Large shapes through ONE build of the
library (TV_HIP_SO selects it; tv_set_attn_bwd_mask launches the backward kernels one at a time).  GPU box, diagnostic."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "deepl-project_amd"))
import torch
from transvae.hip import ops, _lib as L
dev = torch.device("cuda:0")
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 64
shapes = ((384, 4096, 6), (768, 1024, 8), (1536, 256, 12)) if len(sys.argv) <= 2 else ((384, 4096, 6),)
lib = L.load()
lib.tv_set_attn_bwd_mask.argtypes = [ctypes.c_int]
def tm(fn, it=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
tot = 0.0
line = []
for (C, N, nblk) in shapes:
    heads = C // 64
    g = torch.Generator(device=dev).manual_seed(1)
    qkv = torch.randn(mb, N, 3 * C, device=dev, generator=g).to(torch.bfloat16)
    go = torch.randn(mb, N, C, device=dev, generator=g).to(torch.bfloat16)
    o = torch.empty(mb, N, C, device=dev, dtype=torch.bfloat16)
    lse = torch.empty(mb, heads, N, device=dev)
    delta = torch.empty(2, mb, heads, N, device=dev)
    dqkv = torch.zeros_like(qkv)
    P = ops._p
    fwd = lambda: L.check(lib.tv_attn_fwd(P(qkv), P(o), P(lse), mb, N, heads, 0.125, ops._stream()))
    bwd = lambda: L.check(lib.tv_attn_bwd(P(qkv), P(o), P(go), P(lse), P(delta), None, P(dqkv), mb, N, heads, 0.125, ops._stream()))
    fwd(); bwd()
    tf = tm(fwd)
    ts = {}
    for name, mask in (("dq", 2), ("dkv", 4), ("all", 7)):
        lib.tv_set_attn_bwd_mask(mask); ts[name] = tm(bwd)
    lib.tv_set_attn_bwd_mask(7)
    flop = 4.0 * mb * heads * N * N * 64
    print(f"N={N:5d}: fwd {tf:6.3f} ms {flop/tf/1e9:5.0f} TF/s | dq {ts['dq']:6.3f} ({1.5*flop/ts['dq']/1e9:5.0f}) dkv {ts['dkv']:6.3f} ({2.0*flop/ts['dkv']/1e9:5.0f}) bwd {ts['all']:6.3f} ms", flush=True)
    tot += nblk * (tf + ts["all"])
print(f"attention per micro-batch of {mb}: {tot:.1f} ms   [{os.path.basename(L.SO_PATH)}]")
