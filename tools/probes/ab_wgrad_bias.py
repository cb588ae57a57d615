"""Weight gradient WITH the bias gradient of the layers that stay on the single-tap kernel (linear layers, 3x3 at W < 64),
through ONE build of the library (TV_HIP_SO selects it):  python tools/probes/ab_wgrad_bias.py [mb]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "deepl-project_amd"))
from transvae.hip import ops
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
if os.environ.get("AB_KX3_LINEAR"):     # linear layers through the one-tap instantiation of the kx3 loop: "enable[,ring]"
    v = [int(t) for t in os.environ["AB_KX3_LINEAR"].split(",")]
    L.load().tv_set_wgrad_kx3(v[0], v[1] if len(v) > 1 else 0, 0)
    print("linear layers: one-tap kx3 instantiation", v)
g = torch.Generator(device=dev).manual_seed(0)
tot = {True: 0.0, False: 0.0}
for (hw, Cin, Cout, n) in [(16, 1536, 6144, 24), (16, 6144, 1536, 24), (16, 1536, 4608, 12), (16, 1536, 1536, 12), (32, 768, 3072, 16), (32, 3072, 768, 16),
                           (32, 768, 2304, 8), (64, 384, 1536, 12), (64, 1536, 384, 12), (64, 384, 1152, 6)]:
    M = mb * hw * hw
    x = torch.randn(M, Cin, device=dev, generator=g).to(bf)
    w = torch.randn(Cout, Cin, device=dev, generator=g) * Cin ** -0.5
    gz = torch.randn(M, Cout, device=dev, generator=g).to(bf)
    f = 2.0 * M * Cin * Cout
    geo = ops._Geo("linear", x, w)
    for bias in (False, True):
        t = min(tm(lambda: ops.conv_wgrad(geo, w, x, gz, bias)) for _ in range(3))
        tot[bias] += n * t
        dw, db = ops.conv_wgrad(geo, w, x, gz, bias)
        print(f"linear {Cin:5d}->{Cout:<5d}@{hw:<3d} wgrad{'+bias' if bias else '     '} {t:7.3f} ms {f / t / 1e9:6.0f} TF/s   sum {float(dw.sum()):.6e} {float(db.sum()) if bias else 0.0:.6e}", flush=True)
    del x, w, gz
for (hw, Cc, n) in [(16, 1536, 12), (32, 768, 8)]:
    x = torch.randn(mb, hw, hw, Cc, device=dev, generator=g).to(bf)
    gy = torch.randn(mb, hw, hw, Cc, device=dev, generator=g).to(bf)
    w = torch.zeros(Cc, 3, 3, Cc, device=dev)
    f = 2.0 * mb * hw * hw * 9 * Cc * Cc
    geo = ops._Geo("c3s1", x, w)
    for bias in (False, True):
        t = min(tm(lambda: ops.conv_wgrad(geo, w, x, gy, bias)) for _ in range(3))
        tot[bias] += n * t
        dw, db = ops.conv_wgrad(geo, w, x, gy, bias)
        print(f"c3s1 {Cc:5d}@{hw:<4d} wgrad{'+bias' if bias else '     '} {t:7.3f} ms {f / t / 1e9:6.0f} TF/s   sum {float(dw.sum()):.6e} {float(db.sum()) if bias else 0.0:.6e}", flush=True)
    del x, gy, w
print(f"TOTAL over the model's instances: no bias {tot[False]:.2f} ms, with bias {tot[True]:.2f} ms per micro-batch of {mb}")
