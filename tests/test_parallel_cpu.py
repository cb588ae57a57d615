"""The N>1 path on CPU: world_size-2 gloo run of transvae.parallel.train_step.

The data-parallel step (shard the global batch by image, accumulate micro-batches under no_sync,
all-reduce on the last one, clip, AdamW) is model agnostic; the TransVAE kernels need a GPU, so a
small torch module with per-sample normalisation stands in for the model here.  Checked: the
2-rank result equals the single-process full-batch step, and only the last micro-batch syncs."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

from transvae.parallel import micro_batches, shard_range, train_step, wrap_ddp


def test_shard_range_covers_batch():
    for gb, w in [(256, 8), (256, 1), (10, 4), (7, 3)]:
        got = [shard_range(gb, w, r) for r in range(w)]
        assert got[0][0] == 0 and sum(c for _, c in got) == gb
        for (s0, c0), (s1, _) in zip(got, got[1:]):
            assert s0 + c0 == s1
    assert shard_range(256, 8, 3) == (96, 32)
    with pytest.raises(ValueError):
        shard_range(8, 2, 2)


def test_micro_batches():
    assert micro_batches(32, 16) == [(0, 16), (16, 16)]
    assert micro_batches(10, 4) == [(0, 4), (4, 4), (8, 2)]
    with pytest.raises(ValueError):
        micro_batches(4, 0)


class Net(nn.Module):
    def __init__(self):
        super().__init__()
        self.c1 = nn.Conv2d(3, 8, 3, padding=1)
        self.gn = nn.GroupNorm(4, 8)      # per-sample statistics, like every norm on the TransVAE path
        self.c2 = nn.Conv2d(8, 3, 3, padding=1)

    def forward(self, x):
        return self.c2(torch.nn.functional.silu(self.gn(self.c1(x))))


def _loss(m, xb):
    return (m(xb) - xb).abs().mean()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, x_np, q):
    x = torch.from_numpy(x_np)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    net = Net()
    ddp = wrap_ddp(net, None)
    assert isinstance(ddp, nn.parallel.DistributedDataParallel)
    opt = torch.optim.AdamW(net.parameters(), lr=1e-2, betas=(0.9, 0.95), weight_decay=0.0)
    calls = {"sync": 0, "nosync": 0}
    orig = ddp.no_sync

    def counting_no_sync():
        calls["nosync"] += 1
        return orig()
    ddp.no_sync = counting_no_sync
    s, c = shard_range(x.shape[0], world, rank)
    for _ in range(2):
        train_step(ddp, opt, x[s:s + c], 2, _loss, grad_clip=1.0, global_batch=x.shape[0])
    if rank == 0:
        q.put(({k: v.numpy().copy() for k, v in net.state_dict().items()}, calls["nosync"]))  # numpy: pickled by value
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_step_equals_single_process():
    torch.manual_seed(1)
    x = torch.rand(12, 3, 8, 8)      # 6 images per rank -> 3 micro-batches of 2
    # single process, full batch in one micro-batch
    torch.manual_seed(0)
    ref = Net()
    opt = torch.optim.AdamW(ref.parameters(), lr=1e-2, betas=(0.9, 0.95), weight_decay=0.0)
    for _ in range(2):
        train_step(ref, opt, x, 12, _loss, grad_clip=1.0, global_batch=12)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, x.numpy(), q)) for r in range(2)]
    for p in procs:
        p.start()
    sd, nosync_calls = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert nosync_calls == 2 * 2          # 3 micro-batches per step: 2 without sync, the last one syncs
    for k, v in ref.state_dict().items():
        assert torch.allclose(torch.from_numpy(sd[k]), v, rtol=1e-4, atol=1e-6), k
