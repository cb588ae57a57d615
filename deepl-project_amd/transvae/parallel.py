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
    """L1 + kl_weight * KL: the closed-form terms of the reference loss
    (R/transvae/losses/vae_loss.py:83-84,94-96).  KL is summed over the latent and averaged over
    the batch; the clamp keeps exp() finite in fp32 (P/.../vae_loss.py:96-102 does the same)."""
    l1 = (recon.float() - x.float()).abs().mean()
    lv = logvar.float().clamp(-30.0, 20.0)
    kl = -0.5 * torch.sum(1 + lv - mu.float().pow(2) - lv.exp()) / x.shape[0]
    return l1 + kl_weight * kl


def wrap_ddp(model: nn.Module, device: Optional[torch.device], bucket_mb: int = 128) -> nn.Module:
    """DDP with buckets sized for xGMI (few large all-reduces; the 1536-wide stages hold 78 % of the
    gradient bytes and finish mid-backward) -- a no-op wrapper when not distributed."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return model
    ids = [device.index] if device is not None and device.type == "cuda" else None
    return nn.parallel.DistributedDataParallel(model, device_ids=ids, bucket_cap_mb=bucket_mb,
                                               gradient_as_bucket_view=True, broadcast_buffers=False,
                                               find_unused_parameters=False)


def _packed_weight_cache(x: torch.Tensor):
    """The weights do not change between the micro-batches of a step: keep their bf16 repacks (HIP path only)."""
    if x.is_cuda:
        from .hip import ops
        return ops.packed_weight_cache()
    return contextlib.nullcontext()


def train_step(ddp_model: nn.Module, optimizer: torch.optim.Optimizer, x_local: torch.Tensor, micro: int,
               forward_loss: Callable[[nn.Module, torch.Tensor], torch.Tensor], grad_clip: float = 1.0,
               global_batch: Optional[int] = None) -> torch.Tensor:
    """One optimizer step over this rank's images.

    forward_loss(model, x_mb) returns the MEAN loss over x_mb.  Each micro-batch loss is weighted by
    count/global_batch * world_size so that, after DDP's gradient averaging, the result equals the
    gradient of the mean loss over the GLOBAL batch -- identical to a single-process full-batch step.
    Returns the (detached) local weighted loss sum; no host sync happens here.
    """
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    n_local = x_local.shape[0]
    if global_batch is None:
        global_batch = n_local * world
    chunks = micro_batches(n_local, micro)
    total = torch.zeros((), device=x_local.device, dtype=torch.float32)
    optimizer.zero_grad(set_to_none=True)
    with _packed_weight_cache(x_local):
        for i, (s, c) in enumerate(chunks):
            last = i == len(chunks) - 1
            sync_ctx = contextlib.nullcontext() if (last or not hasattr(ddp_model, "no_sync")) else ddp_model.no_sync()
            with sync_ctx:
                loss = forward_loss(ddp_model, x_local[s:s + c]) * (c * world / global_batch)
                loss.backward()
            total += loss.detach()
    if grad_clip is not None and grad_clip > 0:
        torch.nn.utils.clip_grad_norm_(ddp_model.parameters(), grad_clip)
    optimizer.step()
    return total
