"""Timing experiment for VERDICT item 5(a): GroupNorm statistics in the producing convolution's epilogue.
Library A = the in-tree build; library B = conv3x3_halo.hip built with -DTV_EXP_GN_EPI (the register epilogue of the plain /
residual forms also accumulates per-channel sum and sum of squares of its rounded outputs, reduce-scatters them over the 16
pixels of a DPP row and writes 192 partial values per wave).  Prints the convolution times with / without and the time of the
standalone statistics pass (tv_gn_stats) it would replace.  Run once per library (TV_HIP_SO), GPU box."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "deepl-project_amd"))
import torch
from transvae.hip import ops, _lib as L
dev = torch.device("cuda:0")
lib = L.load()
bf = torch.bfloat16
mb = 64
def tm(fn, it=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
probe = None
if hasattr(lib, "tv_set_gn_epi_probe"):
    lib.tv_set_gn_epi_probe.argtypes = [ctypes.c_void_p]
    probe = torch.zeros(16384 * 8 * 256 + 1024, device=dev)
    lib.tv_set_gn_epi_probe(probe.data_ptr())
g = torch.Generator(device=dev).manual_seed(0)
for hw in (256, 128):
    C = 192
    x = torch.randn(mb, hw, hw, C, device=dev, generator=g).to(bf)
    w = torch.randn(C, 3, 3, C, device=dev, generator=g) * (9 * C) ** -0.5
    b = torch.randn(C, device=dev, generator=g) * 0.1
    res = torch.randn(mb, hw, hw, C, device=dev, generator=g).to(bf)
    t_plain = min(tm(lambda: ops.conv_forward(x, w, b, None, "c3s1", L.ACT_NONE, False)[0]) for _ in range(3))
    t_res = min(tm(lambda: ops.conv_forward(x, w, b, res, "c3s1", L.ACT_NONE, False)[0]) for _ in range(3))
    stats = torch.empty(mb, C, 2, device=dev)
    part = torch.empty(int(lib.tv_gn_partial_count(mb, hw * hw, C)), device=dev)
    t_stats = min(tm(lambda: L.check(lib.tv_gn_stats(ops._p(x), ops._p(stats), ops._p(part), mb, hw * hw, C, ops._stream()))) for _ in range(3))
    print(f"{os.path.basename(L.SO_PATH):28s} 192 @{hw}: conv plain {t_plain:.3f} ms  conv+res {t_res:.3f} ms  | standalone gn_stats + finalize {t_stats:.3f} ms"
          + ("  [epilogue statistics ON]" if probe is not None else ""), flush=True)
