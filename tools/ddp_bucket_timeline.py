#!/usr/bin/env python3
"""When do DDP's gradient buckets become ready inside a backward pass of TransVAE-Large?  (one GPU, no communication)

    python tools/ddp_bucket_timeline.py [--images 32] [--bucket-mb 128] > profiles/r03_ddp_bucket_timeline.json

Wraps the model in DistributedDataParallel (world size 1, gloo rendezvous on 127.0.0.1), replaces the all-reduce by
transvae.parallel.BucketTimeline (stamps a device event per bucket, moves no data) and runs forward + backward of ONE
micro-batch of `--images` images -- 32 is what each rank of an 8-GPU job holds of the global batch of 256.  The output lists
the buckets in the order they became ready: size, first / last parameter, device time since the start of the backward
pass, and the cumulative share of gradient bytes.  It is a single-GPU measurement of bucket READINESS; the all-reduce time
itself needs a multi-GPU node."""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "deepl-project_amd"))
import torch
import torch.distributed as dist


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=32)
    ap.add_argument("--bucket-mb", type=int, default=128)
    ap.add_argument("--variant", default="large")
    args = ap.parse_args()
    import bench
    from transvae import TransVAE
    from transvae.parallel import BucketTimeline, vae_bench_loss, wrap_ddp
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
    dist.init_process_group("gloo", rank=0, world_size=1)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    with torch.device(dev):
        model = TransVAE(variant=args.variant, compression_ratio=16, latent_dim=32, clamp_latent=True)
    bench.init_scaled_(model, seed=0)
    model.train()
    ddp = wrap_ddp(model, dev, bucket_mb=args.bucket_mb, force=True)
    tl = BucketTimeline(ddp)
    x = torch.rand(args.images, 3, 256, 256, device=dev)
    eps = torch.randn(args.images, 32, 16, 16, device=dev)
    rep = None
    for it in range(2):                      # first pass warms up (operand packing, allocator); the second is reported
        recon, mu, logvar = ddp(x, eps=eps)
        loss = vae_bench_loss(recon, x, mu, logvar)
        torch.cuda.synchronize()
        tl.start()
        loss.backward()
        rep = tl.report()
        ddp.zero_grad(set_to_none=True)
    b = rep["buckets"]
    half = next(r for r in b if r["cum_bytes_frac"] >= 0.5)
    p90 = next(r for r in b if r["cum_bytes_frac"] >= 0.9)
    rep["summary"] = {"images": args.images, "bucket_cap_mib": args.bucket_mb, "n_buckets": len(b),
                      "half_of_the_bytes_ready_at_frac_of_backward": half["ready_frac_of_backward"],
                      "90pct_of_the_bytes_ready_at_frac_of_backward": p90["ready_frac_of_backward"],
                      "last_bucket_mib": b[-1]["mib"], "last_bucket_ready_frac": b[-1]["ready_frac_of_backward"],
                      "what": "device time at which each bucket's gradients were complete; no all-reduce ran (world size 1)"}
    sys.stdout.flush()
    print(json.dumps(rep, indent=1))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
