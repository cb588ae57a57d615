"""A/B of the 3x3 / stride-1 weight gradient: kx-triple kernel (csrc/wgrad_kx3.hip, ring 3 / 4) against the single-tap kernel,
same process, interleaved rounds (GPU box):   python tools/probes/ab_wgrad3.py [mb]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "deepl-project_amd"))
from transvae.hip import ops
from transvae.hip import _lib as L
dev = torch.device("cuda:0")
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 64
BIAS = os.environ.get("AB_BIAS", "1") == "1"      # with the bias gradient (the model's convolutions all have one)
bf = torch.bfloat16
lib = L.load()


def tm(fn, it=10):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it


g = torch.Generator(device=dev).manual_seed(0)
variants = [("single", (0, 0, 0)), ("kx3 r4", (1, 4, 0)), ("kx3 r3", (1, 3, 0))] + [(f"kx3 r4 v{v}", (1, 4 + 10 * v, 0)) for v in (1, 2, 3, 4)]
variants += [("kx3 t192", (1, 0, 200000))]
if len(sys.argv) > 2:
    variants += [(f"kx3 b{b}", (1, 0, int(b))) for b in sys.argv[2].split(",")]
for (hw, Cc) in [(256, 192), (128, 192), (64, 384), (32, 768), (16, 1536)]:
    x = torch.randn(mb, hw, hw, Cc, device=dev, generator=g).to(bf)
    gy = torch.randn(mb, hw, hw, Cc, device=dev, generator=g).to(bf)
    w = torch.zeros(Cc, 3, 3, Cc, device=dev)
    f = 2.0 * mb * hw * hw * 9 * Cc * Cc
    geo = ops._Geo("c3s1", x, w)
    res = {}
    ref = None
    for rnd in range(3):
        for name, cfg in variants:
            lib.tv_set_wgrad_kx3(*cfg)
            t = tm(lambda: ops.conv_wgrad(geo, w, x, gy, BIAS))
            res.setdefault(name, []).append(t)
            if rnd == 0:
                dw, db = ops.conv_wgrad(geo, w, x, gy, BIAS)
                if ref is None:
                    ref = (dw.clone(), db.clone() if BIAS else None)
                else:
                    e = float((dw - ref[0]).norm() / ref[0].norm()), (float((db - ref[1]).norm() / ref[1].norm()) if BIAS else 0.0)
                    assert e[0] < 1e-4 and e[1] < 1e-4, (name, e)
    lib.tv_set_wgrad_kx3(2, 0, 0)
    for name, ts in res.items():
        t = min(ts)
        print(f"wgrad{' +bias' if BIAS else '      '} c3s1 {Cc:4d}@{hw:<4d} mb{mb} {name:12s} min {t:7.3f} ms {f / t / 1e9:6.0f} TF/s   median {sorted(ts)[len(ts) // 2]:7.3f}", flush=True)
    del x, gy, w
