"""Time the 3x3 / stride-1 convolution layers of the model (forward and data gradient) through ONE build of the library
(TV_HIP_SO selects it) and print a checksum of every result, so that two builds can be compared line by line:

    python tools/probes/ab_lib.py                       # the in-tree build
    TV_HIP_SO=tools/probes/abl/lib_pp.so python tools/probes/ab_lib.py

GPU box.  mb = micro-batch (default 64)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "deepl-project_amd"))
from transvae.hip import ops
from transvae.hip import _lib as L
dev = torch.device("cuda:0")
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 64
quick = len(sys.argv) > 2 and sys.argv[2] == "quick"     # only the 192-channel layers, forward and data gradient
wide = len(sys.argv) > 2 and sys.argv[2] == "wide"       # only the 384 / 768 / 1536-channel layers, forward and data gradient
bf = torch.bfloat16


def tm(fn, it=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it


linear = len(sys.argv) > 2 and sys.argv[2] == "linear"   # the linear layers of the three transformer widths
print("library:", L.SO_PATH)
if os.environ.get("TV_AB_HALO"):
    L.load().tv_set_igemm_halo(int(os.environ["TV_AB_HALO"]))
if os.environ.get("TV_AB_PERSIST") is not None:    # 0 = one tile per block, 1 = heuristic column walk, n = forced walk
    L.load().tv_set_igemm_persist(int(os.environ["TV_AB_PERSIST"]))
if os.environ.get("TV_AB_CFG"):      # bm,bn,stages,bk
    L.load().tv_set_igemm_config(*[int(v) for v in os.environ["TV_AB_CFG"].split(",")])
g = torch.Generator(device=dev).manual_seed(0)
if linear:
    for (hw, Cin, Cout) in [(16, 1536, 6144), (16, 6144, 1536), (16, 1536, 4608), (32, 768, 3072), (32, 3072, 768), (64, 384, 1536), (64, 1536, 384)]:
        M = mb * hw * hw
        x = torch.randn(M, Cin, device=dev, generator=g).to(bf)
        w = torch.randn(Cout, Cin, device=dev, generator=g) * Cin ** -0.5
        b = torch.randn(Cout, device=dev, generator=g) * 0.1
        res = torch.randn(M, Cout, device=dev, generator=g).to(bf)
        gz = torch.randn(M, Cout, device=dev, generator=g).to(bf)
        f = 2.0 * M * Cin * Cout
        geo = ops._Geo("linear", x, w)
        cases = [("fwd", lambda: ops.conv_forward(x, w, b, None, "linear", L.ACT_NONE, False)[0]),
                 ("fwd gelu", lambda: ops.conv_forward(x, w, b, None, "linear", L.ACT_GELU, "deriv")[0]),
                 ("fwd+res", lambda: ops.conv_forward(x, w, b, res, "linear", L.ACT_NONE, False)[0]),
                 ("dgrad", lambda: ops.conv_dgrad(geo, w, gz, x.shape)),
                 ("wgrad", lambda: ops.conv_wgrad(geo, w, x, gz, True)[0])]
        for name, fn in cases:
            t = min(tm(fn), tm(fn))
            y = fn().float()
            print(f"linear {Cin:5d}->{Cout:<5d}@{hw:<3d} {name:10s} {t:7.3f} ms {f / t / 1e9:6.0f} TF/s   sum {float(y.sum()):.6e} abs {float(y.abs().sum()):.6e}", flush=True)
        del x, w, res, gz
    sys.exit(0)
for (hw, Cc) in ([(256, 192), (128, 192)] if quick else ([(64, 384), (32, 768), (16, 1536)] if wide else [(256, 192), (128, 192), (64, 384), (32, 768), (16, 1536)])):
    x = torch.randn(mb, hw, hw, Cc, device=dev, generator=g).to(bf)
    w = torch.randn(Cc, 3, 3, Cc, device=dev, generator=g) * (9 * Cc) ** -0.5
    b = torch.randn(Cc, device=dev, generator=g) * 0.1
    res = torch.randn(mb, hw, hw, Cc, device=dev, generator=g).to(bf)
    f = 2.0 * mb * hw * hw * 9 * Cc * Cc
    geo = ops._Geo("c3s1", x, w)
    cases = [("fwd", lambda: ops.conv_forward(x, w, b, None, "c3s1", L.ACT_NONE, False)[0]),
             ("fwd+res", lambda: ops.conv_forward(x, w, b, res, "c3s1", L.ACT_NONE, False)[0]),
             ("fwd gelu", lambda: ops.conv_forward(x, w, b, None, "c3s1", L.ACT_GELU, "deriv")[0]),
             ("dgrad", lambda: ops.conv_dgrad(geo, w, res, x.shape)),
             ("dgrad*deriv", lambda: ops.conv_dgrad(geo, w, res, x.shape, None, x, L.ACT_DERIV))]
    for name, fn in (cases[:1] + cases[3:4] if (quick or wide) else cases):
        t = min(tm(fn), tm(fn))
        y = fn().float()
        print(f"c3s1 {Cc:5d}@{hw:<4d} {name:12s} {t:7.3f} ms {f / t / 1e9:6.0f} TF/s   sum {float(y.sum()):.6e} abs {float(y.abs().sum()):.6e}", flush=True)
    del x, w, res
