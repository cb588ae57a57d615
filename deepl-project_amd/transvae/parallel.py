"""Data-parallel train step around the TransVAE path (one process per GPU, RCCL over xGMI).

The path shards by image (no BatchNorm; GroupNorm / RMSNorm / LayerNorm / attention are per
sample -- SURVEY.md section 8e), so multi-GPU training is plain data parallelism with ONE exchange
per optimizer step: the gradient all-reduce.  This module mirrors the caller pattern of
R/train.py:557-646 with its two communication defects removed:

  * the reference wraps the model in DDP but never uses ``no_sync()`` while it accumulates 4
    micro-batches, so it all-reduces 4.2 GB of gradients four times per step;
    here only the LAST micro-batch of a step synchronises (buckets overlap with its backward)
  * no per-step ``.item()`` host syncs.

`torch.nn.parallel.DistributedDataParallel` (the reference's own wrapper, R/train.py:672-674) is
kept as the reducer: backend "nccl" is RCCL on ROCm; on CPU the same code runs over gloo (tests).
"""
from __future__ import annotations

import contextlib
from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist
import torch.nn as nn


def shard_range(global_batch: int, world_size: int, rank: int) -> Tuple[int, int]:
    """(start, count) of this rank's slice of a global batch; remainders go to the low ranks."""
    if world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError(f"bad rank/world_size {rank}/{world_size}")
    base, rem = divmod(global_batch, world_size)
    count = base + (1 if rank < rem else 0)
    start = rank * base + min(rank, rem)
    return start, count


def micro_batches(n: int, micro: int):
    """[(start, count)] covering n samples in chunks of at most `micro`."""
    if micro <= 0:
        raise ValueError("micro batch must be positive")
    return [(s, min(micro, n - s)) for s in range(0, n, micro)]


def vae_bench_loss(recon: torch.Tensor, x: torch.Tensor, mu: torch.Tensor, logvar: torch.Tensor,
                   kl_weight: float = 1e-8) -> torch.Tensor:
    """L1 + kl_weight * KL: the closed-form terms of the reference loss, in fp32.

    R/transvae/losses/vae_loss.py:83-84 (L1 = mean |recon - target|) and :94-96 (KL summed over everything and divided
    by batch * H_lat * W_lat, i.e. summed over the latent channels and averaged over batch and latent pixels).  logvar is
    clamped to [-30, 20] first, as the bf16 trainer does before it calls the loss (R/train_2.py:316-318)."""
    if recon.is_cuda:      # one HIP pass: value and the three gradients
        from .losses.vae_loss import fused_l1_kl
        return fused_l1_kl(recon, x, mu, logvar, 1.0, kl_weight, logvar_clip=(-30.0, 20.0))[2]
    l1 = (recon.float() - x.float()).abs().mean()
    mu32 = mu.float()
    lv = logvar.float().clamp(-30.0, 20.0)
    kl = -0.5 * torch.sum(1 + lv - mu32.pow(2) - lv.exp()) / (mu.shape[0] * mu.shape[2] * mu.shape[3])
    return l1 + kl_weight * kl


def warmup_lr(base_lr: float, step: int, warmup_steps: int) -> float:
    """Linear warm-up then constant: the LambdaLR of R/train_2.py:266-273 (factor step / warmup_steps while
    step < warmup_steps, so the very first optimizer step runs at lr 0)."""
    if warmup_steps > 0 and step < warmup_steps:
        return base_lr * float(step) / float(max(1, warmup_steps))
    return base_lr


def wrap_ddp(model: nn.Module, device: Optional[torch.device], bucket_mb: int = 128, grad_exchange: str = "fp32",
             force: bool = False) -> nn.Module:
    """DDP with buckets sized for xGMI (few large all-reduces; the 1536-wide stages hold 78 % of the
    gradient bytes and finish mid-backward) -- a no-op wrapper when not distributed.

    grad_exchange "bf16": the buckets are rounded to bf16 for the all-reduce and widened back (DDP's
    bf16_compress_hook): 2.1 GB instead of 4.2 GB per step over xGMI.  The gradients stay fp32 in `.grad`; what changes is
    one rounding of each rank's bucket before the sum (relative 2^-9 per element).
    force: wrap even with a single rank (bucket-timeline measurements on one GPU)."""
    if grad_exchange not in ("fp32", "bf16"):
        raise ValueError(f"unknown gradient exchange {grad_exchange!r}")
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not force):
        return model
    ids = [device.index] if device is not None and device.type == "cuda" else None
    ddp = nn.parallel.DistributedDataParallel(model, device_ids=ids, bucket_cap_mb=bucket_mb,
                                              gradient_as_bucket_view=True, broadcast_buffers=False,
                                              find_unused_parameters=False)
    if grad_exchange == "bf16":
        from torch.distributed.algorithms.ddp_comm_hooks import default_hooks
        ddp.register_comm_hook(None, default_hooks.bf16_compress_hook)
    return ddp


class BucketTimeline:
    """Records WHEN each DDP gradient bucket becomes ready inside a backward pass (device time), without communicating:
    a comm hook that stamps an event on the stream the gradients were produced on and hands the bucket back unchanged.
    Evidence for the overlap claim of SURVEY section 5 (the bulk of the gradient bytes is ready mid-backward) on ONE GPU;
    what it cannot show is the all-reduce itself.  Usage: tl = BucketTimeline(ddp); tl.start(); loss.backward(); tl.report()."""

    def __init__(self, ddp: nn.Module):
        self.records = []
        self.t0 = None
        self.names = {id(p): n for n, p in ddp.module.named_parameters()}
        ddp.register_comm_hook(self, BucketTimeline._hook)

    @staticmethod
    def _hook(self, bucket):
        import time
        buf = bucket.buffer()
        self.cuda = buf.is_cuda
        if buf.is_cuda:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
        else:
            ev = time.perf_counter()          # (CPU tensors: host time; the tests' stand-in model)
        params = bucket.parameters()
        self.records.append((bucket.index(), buf.numel() * buf.element_size(), [self.names.get(id(p), "?") for p in params], ev))
        fut = torch.futures.Future()
        fut.set_result(buf)
        return fut

    def start(self, cuda: bool = True):
        import time
        self.records.clear()
        self.cuda = cuda and torch.cuda.is_available()
        if self.cuda:
            self.t0 = torch.cuda.Event(enable_timing=True)
            self.t0.record()
        else:
            self.t0 = time.perf_counter()

    def _since(self, ev):
        return self.t0.elapsed_time(ev) if self.cuda else (ev - self.t0) * 1e3

    def report(self):
        import time
        if self.cuda:
            end = torch.cuda.Event(enable_timing=True)
            end.record()
            torch.cuda.synchronize()
        else:
            end = time.perf_counter()
        total_ms = self._since(end)
        total_bytes = sum(r[1] for r in self.records)
        out, cum = [], 0
        for idx, nbytes, names, ev in self.records:          # (in the order the buckets became ready)
            cum += nbytes
            t = self._since(ev)
            out.append({"bucket": idx, "mib": round(nbytes / 2**20, 1), "params": len(names), "first": names[0], "last": names[-1],
                        "ready_ms": round(t, 2), "ready_frac_of_backward": round(t / total_ms, 4),
                        "cum_bytes_frac": round(cum / total_bytes, 4)})
        return {"backward_ms": round(total_ms, 2), "gradient_bytes": total_bytes, "buckets": out}


def _accumulate_in_place(x: torch.Tensor, on: bool):
    """Micro-batches after the first add their weight gradients straight into param.grad (HIP path; see ops)."""
    if x.is_cuda and on:
        from .hip import ops
        return ops.accumulate_grads_in_place(True)
    return contextlib.nullcontext()


_NO_SWAP = __import__("os").environ.get("TV_NO_GRAD_SWAP") == "1"      # A/B hook (bench.py --allow-tuning-env)
swap_stats = {"micro_batches": 0, "tensors": 0}     # diagnostics (tests): micro-batches that ran with swapped accumulators


def _swap_out_autograd_grads(params, x: torch.Tensor, in_place: bool):
    """On an in-place micro-batch the kernels add the weight / bias gradients of the GEMM-shaped layers straight into
    `param.grad`; every OTHER parameter (norm affines, folded projections, DC weights, ...: ~430 tensors of Large) still gets
    its gradient from autograd, whose AccumulateGrad node launches one small add per tensor and micro-batch (1282 launches
    per optimizer step, tools/probes/launch_census.py).  Instead: take those accumulators out (`grad = None`, so autograd
    simply STORES this micro-batch's gradient) and add all of them back with one multi-tensor add afterwards.  Which
    parameters the kernels handle is learned from the kernels themselves (ops.in_place_params, filled during the first
    in-place micro-batch of the process, which therefore still runs the old way)."""
    if not (in_place and x.is_cuda) or _NO_SWAP:
        return None
    from .hip import ops
    known = ops.in_place_params
    if not known:
        return None
    swapped = [(p, p.grad) for p in params if p.grad is not None and id(p) not in known]
    for p, _ in swapped:
        p.grad = None
    swap_stats["micro_batches"] += 1
    swap_stats["tensors"] += len(swapped)
    return swapped


def _swap_in_and_add(swapped):
    if not swapped:
        return
    acc, new = [], []
    for p, g0 in swapped:
        g1 = p.grad
        p.grad = g0
        if g1 is not None:
            acc.append(g0)
            new.append(g1 if g1.dtype == g0.dtype else g1.to(g0.dtype))
    if acc:
        torch._foreach_add_(acc, new)


def _defer_chain(x: torch.Tensor, on: bool):
    if x.is_cuda and on:
        from .hip import ops
        return ops.defer_chain_grads(True)
    return contextlib.nullcontext()


def _packed_weight_cache(x: torch.Tensor):
    """The weights do not change between the micro-batches of a step: keep their bf16 repacks (HIP path only)."""
    if x.is_cuda:
        from .hip import ops
        return ops.packed_weight_cache()
    return contextlib.nullcontext()


def global_grad_norm(params) -> torch.Tensor:
    """L2 norm of all gradients as ONE device scalar (fp32), no host sync."""
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return torch.zeros(())
    norms = torch._foreach_norm(grads, 2.0)
    return torch.linalg.vector_norm(torch.stack([n.float() for n in norms]), 2.0)


def clip_and_step(params, optimizer: torch.optim.Optimizer, grad_clip: Optional[float], counters: Optional[dict] = None):
    """Global-norm clip (R/train.py:610-612) + optimizer step, with the reference trainer's non-finite guard
    (R/train_2.py:328-338: a non-finite loss skips the step; a non-finite loss or activation gradient makes the gradient
    norm non-finite, which is what is tested here) -- all on the device, no host sync:

      * optimizers that accept the GradScaler protocol (`_step_supports_amp_scaling`: torch's fused AdamW, and
        transvae.optim.FusedAdamW) receive `grad_scale = 1 / clip_coef` and `found_inf`: the un-scale IS the clip, done
        inside the update kernel (no separate pass over the 4.2 GB of gradients), and the whole update is skipped when
        found_inf is set;
      * any other optimizer (CPU tests): gradients are multiplied by the coefficient, non-finite ones zeroed, and the
        step is skipped through a host check.

    counters (optional dict of device scalars): 'skipped' is incremented on a skipped step, 'grad_norm' holds the last norm.
    Returns the gradient norm before clipping."""
    params = [p for p in params if p.grad is not None]
    if not params:
        return None
    fused_norm = getattr(optimizer, "fused_clip_step", None)
    if fused_norm is not None:           # transvae.optim.FusedAdamW: norm, clip, guard and update in its own kernels
        norm, found_inf = fused_norm(grad_clip)
    else:
        norm = global_grad_norm(params)
        found_inf = (~torch.isfinite(norm)).float()
        if grad_clip is not None and grad_clip > 0:
            coef = torch.clamp(grad_clip / (norm + 1e-6), max=1.0)
        else:
            coef = torch.ones_like(norm)
        coef = torch.where(found_inf > 0, torch.ones_like(coef), coef)
        if getattr(optimizer, "_step_supports_amp_scaling", False):
            optimizer.grad_scale = (1.0 / coef).reshape(())
            optimizer.found_inf = found_inf.reshape(())
            try:
                optimizer.step()
            finally:
                optimizer.grad_scale = None
                optimizer.found_inf = None
        else:
            if bool(found_inf):      # (host check: this branch is the CPU / unfused path)
                optimizer.zero_grad(set_to_none=True)
            else:
                torch._foreach_mul_([p.grad for p in params], coef)
                optimizer.step()
    if counters is not None:
        counters["skipped"] = counters.get("skipped", 0) + found_inf.reshape(())
        counters["grad_norm"] = norm
    return norm


def train_step(ddp_model: nn.Module, optimizer: torch.optim.Optimizer, x_local: torch.Tensor, micro: int,
               forward_loss: Callable[[nn.Module, torch.Tensor], torch.Tensor], grad_clip: float = 1.0,
               global_batch: Optional[int] = None, counters: Optional[dict] = None) -> torch.Tensor:
    """One optimizer step over this rank's images.

    forward_loss(model, x_mb) returns the MEAN loss over x_mb.  Each micro-batch loss is weighted by
    count/global_batch * world_size so that, after DDP's gradient averaging, the result equals the
    gradient of the mean loss over the GLOBAL batch -- identical to a single-process full-batch step.
    Returns the (detached) local weighted loss sum; no host sync happens here.  A step whose gradients are not
    finite is skipped on the device (see clip_and_step); `counters['skipped']` counts them.
    """
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    n_local = x_local.shape[0]
    if global_batch is None:
        global_batch = n_local * world
    chunks = micro_batches(n_local, micro)
    total = torch.zeros((), device=x_local.device, dtype=torch.float32)
    optimizer.zero_grad(set_to_none=True)
    params = list(ddp_model.parameters())
    with _packed_weight_cache(x_local):
        for i, (s, c) in enumerate(chunks):
            last = i == len(chunks) - 1
            sync_ctx = contextlib.nullcontext() if (last or not hasattr(ddp_model, "no_sync")) else ddp_model.no_sync()
            # in-place accumulation hands autograd no gradient for those parameters, so DDP's hooks would not fire: never on
            # the micro-batch that synchronises (also with ONE rank when the model is DDP-wrapped: wrap_ddp(force=True))
            in_place = i > 0 and (not hasattr(ddp_model, "no_sync") or not last)
            swapped = _swap_out_autograd_grads(params, x_local, in_place)
            # (the Conv-FFN composite gradients wait for the end of the step on every pass DDP does not reduce in)
            with sync_ctx, _accumulate_in_place(x_local, in_place), _defer_chain(x_local, not hasattr(ddp_model, "no_sync") or not last):
                loss = forward_loss(ddp_model, x_local[s:s + c]) * (c * world / global_batch)
                loss.backward()
            _swap_in_and_add(swapped)
            total += loss.detach()
    clip_and_step(params, optimizer, grad_clip, counters)
    return total
