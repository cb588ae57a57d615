"""A/B of the compile-time epilogue forms (tv_set_igemm_epilogue 1 / 0) on the layers that carry a residual add or an
activation gradient (from the saved pre-activation: run-time form either way; from the saved derivative: EPI 2): time of each form and bit-equality of the two results.  GPU box."""
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


def ab(name, flop, fn):
    r, t = [None, None], [1e30, 1e30]
    for on in (1, 0, 1, 0):      # (best of two per form: the first timing after a shape change runs cold)
        lib.tv_set_igemm_epilogue(on)
        r[1 - on] = fn().clone(); t[1 - on] = min(t[1 - on], tm(fn))
    lib.tv_set_igemm_epilogue(1)
    same = torch.equal(r[0], r[1])
    print(f"{name:44s} compile-time {t[0]:7.3f} ms ({flop/t[0]/1e9:5.0f} TF/s) | run-time {t[1]:7.3f} ms ({flop/t[1]/1e9:5.0f} TF/s) | "
          f"{(t[1]/t[0]-1)*100:+5.1f}% | bit-equal {same}", flush=True)
    assert same, name


mb = int(sys.argv[1]) if len(sys.argv) > 1 else 64
bf = torch.bfloat16
# linear layers: forward with residual (ffn_out / proj), data gradient with GELU' (the layer behind a GELU), with residual gradient
for (hw, Cin, Cout) in [(16, 6144, 1536), (32, 3072, 768), (64, 1536, 384), (16, 1536, 1536), (64, 384, 384)]:
    M = mb * hw * hw
    x = torch.randn(M, Cin, device=dev).to(bf)
    w = torch.randn(Cout, Cin, device=dev) * Cin ** -0.5
    b = torch.randn(Cout, device=dev) * 0.1
    res = torch.randn(M, Cout, device=dev).to(bf)
    f = 2.0 * M * Cin * Cout
    ab(f"linear {Cin}->{Cout} @{hw} fwd +residual", f, lambda: ops.conv_forward(x, w, b, res, "linear", L.ACT_NONE, False)[0])
    g = ops._Geo("linear", x, w)
    gz = torch.randn(M, Cout, device=dev).to(bf)
    aux = torch.randn(M, Cin, device=dev).to(bf)
    res_in = torch.randn(M, Cin, device=dev).to(bf)
    ab(f"linear {Cin}->{Cout} @{hw} dgrad *GELU'(pre)", f, lambda: ops.conv_dgrad(g, w, gz, x.shape, None, aux, L.ACT_GELU))
    ab(f"linear {Cin}->{Cout} @{hw} dgrad *deriv", f, lambda: ops.conv_dgrad(g, w, gz, x.shape, None, aux, L.ACT_DERIV))
    ab(f"linear {Cin}->{Cout} @{hw} dgrad +res *deriv", f, lambda: ops.conv_dgrad(g, w, gz, x.shape, res_in, aux, L.ACT_DERIV))
    ab(f"linear {Cin}->{Cout} @{hw} dgrad +res", f, lambda: ops.conv_dgrad(g, w, gz, x.shape, res_in, None, 0))
    del x, w, res, gz, aux, res_in
# 3x3 convolutions: ResBlock conv2 (+residual), data gradient with SiLU' (in front of GroupNorm: none) -- the FFN's 3x3 has GELU
for (hw, Cc) in [(256, 192), (128, 192), (16, 1536), (32, 768), (64, 384)]:
    x = torch.randn(mb, hw, hw, Cc, device=dev).to(bf)
    w = torch.randn(Cc, 3, 3, Cc, device=dev) * (9 * Cc) ** -0.5
    b = torch.randn(Cc, device=dev) * 0.1
    res = torch.randn(mb, hw, hw, Cc, device=dev).to(bf)
    f = 2.0 * mb * hw * hw * 9 * Cc * Cc
    ab(f"c3s1 {Cc}@{hw} fwd +residual", f, lambda: ops.conv_forward(x, w, b, res, "c3s1", L.ACT_NONE, False)[0])
    g = ops._Geo("c3s1", x, w)
    ab(f"c3s1 {Cc}@{hw} dgrad *deriv", f, lambda: ops.conv_dgrad(g, w, res, x.shape, None, x, L.ACT_DERIV))
    ab(f"c3s1 {Cc}@{hw} dgrad +res", f, lambda: ops.conv_dgrad(g, w, res, x.shape, x, None, 0))
    del x, w, res
