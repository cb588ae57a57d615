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


def _worker(rank, world, port, x_np, q, grad_exchange="fp32"):
    x = torch.from_numpy(x_np)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    net = Net()
    ddp = wrap_ddp(net, None, grad_exchange=grad_exchange)
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


def test_two_rank_gloo_bf16_gradient_exchange_stays_within_bf16_of_the_fp32_exchange():
    """wrap_ddp(grad_exchange="bf16"): the buckets cross the wire as bf16 (half the bytes of the 4.2 GB fp32 exchange, SURVEY
    section 5).  One step on two ranks with each exchange: the parameters after the step agree to what one bf16 rounding
    of the summed gradients can move an AdamW update (lr 1e-2: the update is +-lr at most, its relative change ~2^-8)."""
    torch.manual_seed(1)
    x = torch.rand(12, 3, 8, 8)
    ctx = mp.get_context("spawn")
    out = {}
    for mode in ("fp32", "bf16"):
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, 2, port, x.numpy(), q, mode)) for r in range(2)]
        for p in procs:
            p.start()
        out[mode], _ = q.get(timeout=120)
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
    differ = 0
    for k in out["fp32"]:
        a, b = torch.from_numpy(out["fp32"][k]), torch.from_numpy(out["bf16"][k])
        assert torch.allclose(a, b, rtol=0, atol=2 * 2e-2 * 2.0 ** -7), k      # two steps of lr 1e-2, direction changed by <= 2^-7
        differ += int(not torch.equal(a, b))
    assert differ > 0          # the hook did run (the exchange WAS rounded)


def test_wrap_ddp_rejects_unknown_exchange():
    with pytest.raises(ValueError):
        wrap_ddp(Net(), None, grad_exchange="fp8")


# ---- numerical guards of the train step (R/train_2.py:266-273, 316-318, 328-338; R/train.py:610-612) ------------
def test_warmup_lr_follows_reference_lambda():
    from transvae.parallel import warmup_lr
    assert warmup_lr(1e-4, 0, 1000) == 0.0                      # LambdaLR factor step / warmup_steps at step 0
    assert abs(warmup_lr(1e-4, 250, 1000) - 2.5e-5) < 1e-12
    assert warmup_lr(1e-4, 1000, 1000) == 1e-4 and warmup_lr(1e-4, 5000, 1000) == 1e-4
    assert warmup_lr(1e-4, 0, 0) == 1e-4                        # no warm-up


def test_bench_loss_is_the_reference_formula():
    """vae_loss.py:83-84,94-96 written out: L1 mean + 1e-8 * (-0.5 * sum(...)) / (B * H * W); equal to the oracle's."""
    from oracle import transvae_oracle as O
    from transvae.parallel import vae_bench_loss
    g = torch.Generator().manual_seed(0)
    recon, x = torch.randn(3, 3, 16, 16, generator=g), torch.rand(3, 3, 16, 16, generator=g)
    mu, logvar = torch.randn(3, 4, 2, 2, generator=g) * 3, torch.randn(3, 4, 2, 2, generator=g) * 2
    kl = -0.5 * torch.sum(1 + logvar - mu.pow(2) - logvar.exp())
    kl = kl / (mu.shape[0] * mu.shape[2] * mu.shape[3])
    want = torch.nn.functional.l1_loss(recon, x) + 1e-8 * kl
    assert torch.allclose(vae_bench_loss(recon, x, mu, logvar), want, rtol=1e-6)
    assert torch.allclose(O.bench_loss(recon, x, mu, logvar), want, rtol=1e-6)
    # the clamp of train_2.py:316-318 keeps exp() finite
    big = torch.full_like(logvar, 500.0)
    assert torch.isfinite(vae_bench_loss(recon, x, mu, big)) and torch.isfinite(O.bench_loss(recon, x, mu, big, clamp_logvar=True))
    assert not torch.isfinite(O.bench_loss(recon, x, mu, big))


def test_clip_and_step_matches_torch_clip_and_skips_non_finite():
    from transvae.parallel import clip_and_step
    torch.manual_seed(0)
    a, b = Net(), Net()
    b.load_state_dict(a.state_dict())
    x = torch.rand(4, 3, 8, 8)
    oa = torch.optim.AdamW(a.parameters(), lr=1e-2, betas=(0.9, 0.95), weight_decay=0.0)
    ob = torch.optim.AdamW(b.parameters(), lr=1e-2, betas=(0.9, 0.95), weight_decay=0.0)
    (_loss(a, x) * 50).backward()
    (_loss(b, x) * 50).backward()
    counters = {}
    n = clip_and_step(list(a.parameters()), oa, 1.0, counters)
    n_ref = torch.nn.utils.clip_grad_norm_(b.parameters(), 1.0)
    ob.step()
    assert float(n) > 1.0 and abs(float(n) - float(n_ref)) < 1e-5 * float(n_ref)
    for pa, pb in zip(a.parameters(), b.parameters()):
        assert torch.allclose(pa, pb, rtol=1e-5, atol=1e-7)
    assert float(counters["skipped"]) == 0
    # a non-finite gradient: the step is skipped, parameters and Adam moments are untouched, the next step is normal
    before = {k: v.clone() for k, v in a.state_dict().items()}
    m_before = [oa.state[p]["exp_avg"].clone() for p in a.parameters()]
    oa.zero_grad()
    _loss(a, x).backward()
    next(a.parameters()).grad[0, 0, 0, 0] = float("nan")
    clip_and_step(list(a.parameters()), oa, 1.0, counters)
    assert float(counters["skipped"]) == 1
    for k, v in a.state_dict().items():
        assert torch.equal(v, before[k]), k
    for p, m0 in zip(a.parameters(), m_before):
        assert torch.equal(oa.state[p]["exp_avg"], m0)
    oa.zero_grad()
    _loss(a, x).backward()
    clip_and_step(list(a.parameters()), oa, 1.0, counters)
    assert float(counters["skipped"]) == 1 and any(not torch.equal(v, before[k]) for k, v in a.state_dict().items())
