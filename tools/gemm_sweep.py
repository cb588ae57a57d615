"""Per-layer-shape timing of the implicit-GEMM kernels for one TransVAE variant (GPU box).

    python tools/gemm_sweep.py [--mb 32] [--variant large] [--res 256]

Enumerates every conv / linear instance of the model (forward, data-gradient, weight-gradient),
times each distinct shape with HIP events and prints TFLOP/s, the time share and the total, so the
kernel work can be aimed at the shapes that matter.  Diagnostic only.
"""
import argparse
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "deepl-project_amd"))
from transvae.hip import ops  # noqa: E402
from transvae.models.transvae import VARIANT_CONFIGS  # noqa: E402

BF = torch.bfloat16


def layer_list(variant, res, mb):
    """[(name, mode, B, H, W, Cin, Cout, count)] of every GEMM-shaped layer (forward geometry)."""
    cfg = VARIANT_CONFIGS[f"{variant}_f16d32"]
    depths, dims = cfg["depths"], cfg["base_dims"]
    L = collections.OrderedDict()

    def add(name, mode, H, Cin, Cout, n=1):
        key = (name, mode, mb, H, H, Cin, Cout)
        L[key] = L.get(key, 0) + n

    n = len(depths)
    for side in ("enc", "dec"):
        for i in range(n):
            H = res >> i
            d = dims[i]
            if i < 2:
                add(f"res{d}@{H}", "c3s1", H, d, d, 2 * depths[i])
            else:
                T = H
                add(f"qkv{d}@{H}", "linear", T, d, 3 * d, depths[i])
                add(f"proj{d}@{H}", "linear", T, d, d, depths[i])
                add(f"ffn_in{d}@{H}", "linear", T, d, 4 * d, depths[i])
                add(f"ffn_c0{d}@{H}", "linear", T, 4 * d, d, depths[i])
                add(f"ffn_c3x3{d}@{H}", "c3s1", H, d, d, depths[i])
                # round 4: conv.4 (d -> 4d) runs collapsed into the composite Wc = W_out W3 (d -> d): its forward rides in
                # proj_out's launch as a second K source, its data / weight gradients are d -> d launches
                add(f"ffn_wc{d}@{H}", "linear", T, d, d, depths[i])
                add(f"ffn_out{d}@{H}", "linear", T, 4 * d, d, depths[i])
        for i in range(n - 1):
            H = res >> i
            a, b = dims[i], dims[i + 1]
            if side == "enc":
                add(f"down_c1 {a}@{H}", "c3s1", H, a, a)
                add(f"down_c2 {a}->{b}@{H}", "c3s2", H, a, b)
                add(f"down_dc {a}->{b}@{H}", "unshuf", H, a, b)
            else:
                add(f"up_c1 {b}->{a}@{H >> 1}", "c3up", H >> 1, b, a)
                add(f"up_c2 {a}@{H}", "c3s1", H, a, a)
                add(f"up_dc {b}->{a}@{H >> 1}", "shuf", H >> 1, b, 4 * a)
    return L


def time_fn(fn, iters=5):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mb", type=int, default=32)
    ap.add_argument("--variant", default="large")
    ap.add_argument("--res", type=int, default=256)
    ap.add_argument("--cfg", default="", help="igemm tuning: bm,bn,stages,bk (0 = heuristic)")
    ap.add_argument("--brief", action="store_true")
    ap.add_argument("--halo", type=int, default=1, help="3x3 stride-1 halo kernel: 0 off, 1 default, 2 / 3 weight ring depth")
    ap.add_argument("--wring", type=int, default=0, help="wgrad LDS ring depth (0 = heuristic, 2, 3)")
    ap.add_argument("--only", default="", help="substring filter on the layer name")
    ap.add_argument("--wstages", type=int, default=0, help="wgrad pixels per K-step (0 = heuristic, 32, 64)")
    ap.add_argument("--wblocks", type=int, default=0, help="wgrad split-K block target (0 = heuristic)")
    ap.add_argument("--wwaves", type=int, default=0, help="wgrad waves per block (0 = heuristic, 4)")
    args = ap.parse_args()
    if os.environ.get("TV_SWEEP_PERSIST"):   # tv_set_igemm_persist hook (A/B of the loop variants)
        from transvae.hip import _lib as _L
        _L.load().tv_set_igemm_persist(int(os.environ["TV_SWEEP_PERSIST"]))
    if args.cfg:
        from transvae.hip import _lib
        _lib.load().tv_set_igemm_config(*[int(v) for v in args.cfg.split(",")])
    if args.wstages or args.wwaves or args.wblocks:
        from transvae.hip import _lib
        _lib.load().tv_set_wgrad_config(args.wstages, args.wwaves, args.wblocks)
    if args.halo != 1:
        from transvae.hip import _lib
        _lib.load().tv_set_igemm_halo(args.halo)
    if args.wring:
        from transvae.hip import _lib
        _lib.load().tv_set_wgrad_stages(args.wring)
    dev = torch.device("cuda:0")
    rows = []
    for (name, mode, B, H, W, Cin, Cout), count in layer_list(args.variant, args.res, args.mb).items():
        if args.only and args.only not in name:
            continue
        if mode == "linear":
            x = torch.randn(B * H * W, Cin, device=dev).to(BF).requires_grad_(True)
            w = (torch.randn(Cout, Cin, device=dev) * Cin ** -0.5).requires_grad_(True)
            taps, Mout = 1, B * H * W
        else:
            k = {"c3s1": 3, "c3s2": 3, "c3up": 3, "unshuf": 2, "shuf": 1}[mode]
            x = torch.randn(B, H, W, Cin, device=dev).to(BF).requires_grad_(True)
            w = (torch.randn(Cout, k, k, Cin, device=dev) * (k * k * Cin) ** -0.5).requires_grad_(True)
            taps = k * k
            Mout = B * H * W * {"c3s1": 1, "c3s2": 0.25, "c3up": 4, "unshuf": 0.25, "shuf": 1}[mode]
        flop = 2.0 * Mout * Cout * taps * Cin
        y = ops.conv(x, w, None, None, mode=mode) if mode != "linear" else ops.linear(x, w)
        gy = torch.randn_like(y)
        t_f = time_fn(lambda: ops.conv(x, w, None, None, mode=mode) if mode != "linear" else ops.linear(x, w))

        def bwd(need_x, need_w):
            x.requires_grad_(need_x)
            w.requires_grad_(need_w)
            yy = ops.conv(x, w, None, None, mode=mode) if mode != "linear" else ops.linear(x, w)
            return yy

        # backward pieces: time fwd+bwd variants and subtract the forward
        def run_dx():
            bwd(True, False).backward(gy)
            x.grad = None

        def run_dw():
            bwd(False, True).backward(gy)
            w.grad = None
        t_dx = max(time_fn(run_dx) - t_f, 1e-6)
        t_dw = max(time_fn(run_dw) - t_f, 1e-6)
        rows.append((name, mode, count, flop, t_f, t_dx, t_dw))
        del x, w, y, gy
        torch.cuda.empty_cache()
    tot = sum(c * (a + b + d) for _, _, c, _, a, b, d in rows)
    totf = sum(c * 3 * f for _, _, c, f, *_ in rows)
    if args.brief:
        tf = sum(c * a for _, _, c, _, a, b, d in rows)
        tb = sum(c * b for _, _, c, _, a, b, d in rows)
        tw = sum(c * d for _, _, c, _, a, b, d in rows)
        print(f"cfg={args.cfg or 'default':12s} wbkp={args.wstages} ww={args.wwaves} wblk={args.wblocks} fwd {tf:7.1f} ms  dgrad {tb:7.1f} ms  wgrad {tw:7.1f} ms  total {tot:7.1f} ms")
        for name, mode, c, f, a, b, d in rows:
            if name in ("res192@256", "ffn_c3x31536@16", "ffn_in1536@16", "ffn_in384@64", "proj1536@16", "qkv768@32", "ffn_out768@32"):
                print(f"    {name:20s} fwd {f / a / 1e9:5.0f} TF/s   dgrad {f / b / 1e9:5.0f} TF/s   wgrad {f / d / 1e9:5.0f} TF/s")
        return
    print(f"{'layer':28s} {'mode':7s} {'n':>3s} {'GF':>8s} | {'fwd ms':>8s} {'TF/s':>6s} | {'dgrad':>8s} {'TF/s':>6s} | {'wgrad':>8s} {'TF/s':>6s} | share")
    for name, mode, c, f, a, b, d in sorted(rows, key=lambda r: -r[2] * (r[4] + r[5] + r[6])):
        print(f"{name:28s} {mode:7s} {c:3d} {f / 1e9:8.1f} | {a:8.3f} {f / a / 1e9:6.0f} | {b:8.3f} {f / b / 1e9:6.0f} | {d:8.3f} {f / d / 1e9:6.0f} | "
              f"{100 * c * (a + b + d) / tot:5.1f}%")
    print(f"TOTAL {tot:.1f} ms per micro-batch of {args.mb} ({totf / tot / 1e9:.0f} TFLOP/s over all GEMM-shaped work, "
          f"{1e3 * args.mb / tot:.1f} img/s if nothing else ran)")


if __name__ == "__main__":
    main()
