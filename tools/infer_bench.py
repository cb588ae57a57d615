"""Forward-only throughput of the HIP path (encode / decode entry points).  GPU box.

    python tools/infer_bench.py [--variant large] [--res 256] [--batch 64] [--iters 5]

Runs model.encode(), model.decode() and model() under torch.no_grad() -- the fused autograd Functions then save nothing
and skip the pre-activation stores -- and prints images/s and peak memory for each.  Diagnostic; not the headline metric.
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "deepl-project_amd"))
from transvae import TransVAE  # noqa: E402


def timed(fn, iters):
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
    ap.add_argument("--variant", default="large")
    ap.add_argument("--res", type=int, default=256)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--iters", type=int, default=5)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = TransVAE(variant=a.variant, compression_ratio=16, latent_dim=32).to(dev).eval()
    x = torch.rand(a.batch, 3, a.res, a.res, device=dev)
    with torch.no_grad():
        z = model.encode(x)
        z = z[0] if isinstance(z, (tuple, list)) else z
        torch.cuda.reset_peak_memory_stats()
        t_enc = timed(lambda: model.encode(x), a.iters)
        t_dec = timed(lambda: model.decode(z), a.iters)
        t_all = timed(lambda: model(x), a.iters)
    gib = torch.cuda.max_memory_allocated() / 2 ** 30
    print(f"{a.variant} f16d32 @{a.res}x{a.res}, batch {a.batch}, no_grad: encode {1e3 * a.batch / t_enc:8.1f} img/s | "
          f"decode {1e3 * a.batch / t_dec:8.1f} img/s | forward {1e3 * a.batch / t_all:8.1f} img/s | peak {gib:.1f} GiB")


if __name__ == "__main__":
    main()
