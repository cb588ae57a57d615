"""Same-process A/B of the block order of the 8-wave GEMM tiles (tv_set_igemm_supertile): 1 = row-tile-major (round 3),
0 = super-tile chosen per shape, r * 100 + c = forced shape.  Linear layers of TransVAE-Large at micro-batch 64, forward
geometry and the data-gradient geometry (K and N swapped).  Interleaved rounds, median of the per-round averages."""
import os, sys, statistics
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "deepl-project_amd"))
import torch
from transvae.hip import ops, _lib as L
lib = L.load()
lib.tv_set_igemm_supertile.argtypes = [L.C.c_int]
dev = torch.device("cuda:0")
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 64
variants = [1, 0, 408, 804, 506, 605, 216, 1602]
shapes = []
for d, T in ((384, 4096), (768, 1024), (1536, 256)):
    M = mb * T
    shapes += [("qkv%d" % d, M, d, 3 * d), ("ffn_in%d" % d, M, d, 4 * d), ("ffn_c0%d" % d, M, 4 * d, d), ("proj%d" % d, M, d, d),
               ("qkv%d.dgrad" % d, M, 3 * d, d)]
g = torch.Generator(device=dev).manual_seed(0)
print("%-16s %8s %6s %6s | " % ("layer", "M", "K", "N") + " ".join("%9s" % v for v in variants) + "   (ms; TF/s of the best)")
tot = {v: 0.0 for v in variants}
for name, M, K, N in shapes:
    x = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
    w = torch.randn(N, K, device=dev, generator=g) * K ** -0.5
    b = torch.randn(N, device=dev, generator=g) * 0.1
    fn = lambda: ops.conv_forward(x, w, b, None, "linear", L.ACT_NONE, False)[0]
    ref = None
    times = {v: [] for v in variants}
    for rnd in range(5):
        for v in variants:
            lib.tv_set_igemm_supertile(v)
            for _ in range(3):
                y = fn()
            if rnd == 0:
                if ref is None:
                    ref = y.clone()
                else:
                    assert torch.equal(y, ref), (name, v)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record()
            torch.cuda.synchronize()
            times[v].append(e0.elapsed_time(e1) / 20)
    med = {v: statistics.median(times[v]) for v in variants}
    for v in variants:
        tot[v] += med[v]
    best = min(med.values())
    print("%-16s %8d %6d %6d | " % (name, M, K, N) + " ".join("%9.4f" % med[v] for v in variants) + "   %6.0f" % (2.0 * M * K * N / best / 1e9))
    del x, w, b, ref
lib.tv_set_igemm_supertile(0)
print("%-16s %8s %6s %6s | " % ("SUM", "", "", "") + " ".join("%9.4f" % tot[v] for v in variants))
