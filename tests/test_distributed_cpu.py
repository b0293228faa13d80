"""CPU, world_size 2, gloo: the data-parallel exchange steps (segmi.distributed) are correct by
construction — bucketed gradient averaging equals the gradient of the global-batch loss, and the
SyncBN statistic exchange merges to the statistics of the concatenated batch (the reference's
semantics: nn.DataParallel computes the loss on the gathered global batch, base/base_trainer.py:33-38;
utils/sync_batchnorm/batchnorm.py:105-145)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(rank, world, port, fn, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    for p in (ROOT, os.path.join(ROOT, "pytorch-segmentation_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ret[rank] = fn(rank, world)
    finally:
        dist.destroy_process_group()


def _spawn(fn, world=2):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_run, args=(world, _free_port(), fn, ret), nprocs=world, join=True)
    return [ret[r] for r in range(world)]


class _Net(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.a = torch.nn.Conv2d(3, 8, 3, padding=1)
        self.a.weight.data = self.a.weight.data.contiguous(memory_format=torch.channels_last)  # permuted-dense param
        self.b = torch.nn.Linear(8, 5)
        self.unused = torch.nn.Linear(4, 4)      # never produces a gradient: finish() must still reduce its bucket
        self.s = torch.nn.Parameter(torch.ones(()))

    def forward(self, x):
        return self.b(torch.relu(self.a(x)).mean((2, 3))) * self.s


def _grad_worker(rank, world):
    from segmi.distributed import DistributedModel
    torch.manual_seed(100 + rank)          # different init per rank: the wrapper must broadcast rank 0's
    net = _Net()
    dm = DistributedModel(net, bucket_bytes=600)   # tiny buckets -> several collectives
    g = torch.Generator().manual_seed(5)
    X = torch.randn(4 * world, 3, 6, 6, generator=g)
    Y = torch.randn(4 * world, 5, generator=g)
    xs, ys = X[rank * 4:(rank + 1) * 4], Y[rank * 4:(rank + 1) * 4]
    out = {}
    for it, zero in enumerate(("reducer", "set_to_none")):
        if zero == "reducer":
            dm.zero_grad()
        else:
            torch.optim.SGD(net.parameters(), lr=0.1).zero_grad(set_to_none=True)
            dm.reducer.zero_grad()
            for p in net.parameters():
                p.grad = None               # the hook must re-attach bucket views
        loss = ((dm(xs) - ys) ** 2).mean()
        loss.backward()
        dm.finish_gradients()
        out[zero] = {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}
        assert all(p.grad.data_ptr() == dm.reducer._where[id(p)][1].data_ptr() for p in dm.reducer.params if p.grad is not None)
        # parameters without a gradient stay None for the optimizer (reference semantics: no weight decay / momentum on them) ...
        unused_none = net.unused.weight.grad is None and net.unused.bias.grad is None
        # ... while their (zero) bucket slots still went through the collective
        unused_zero = all(float(dm.reducer._where[id(p)][1].abs().max()) == 0.0 for p in net.unused.parameters())
        out[zero]["_unused_ok"] = unused_none and unused_zero
    # reference: one process, global batch
    torch.manual_seed(100)
    ref = _Net()
    ((ref(X) - Y) ** 2).mean().backward()
    refg = {k: p.grad for k, p in ref.named_parameters() if p.grad is not None}
    sd_equal = all(torch.equal(a, b) for a, b in zip(net.state_dict().values(), ref.state_dict().values()))
    return {"nb": len(dm.reducer.buckets), "sd_equal": sd_equal,
            "err": {z: max((out[z][k] - refg[k]).abs().max().item() for k in refg) for z in out},
            "unused_zero": all(out[z]["_unused_ok"] for z in out)}


def test_bucketed_gradient_allreduce_equals_global_batch_gradient():
    res = _spawn(_grad_worker)
    for r in res:
        assert r["nb"] >= 3
        assert r["sd_equal"]
        assert r["unused_zero"]
        for z, e in r["err"].items():
            assert e < 1e-6, (z, e)


def _loss_worker(rank, world):
    """Global-batch loss semantics: shards with DIFFERENT numbers of valid pixels.  Per-rank loss =
    segmi.distributed.global_batch_mean(local sum, local count) -> averaged gradients == gradient of the single-process
    CrossEntropy over the concatenated batch (reference trainer.py:56-66, utils/losses.py:29-31)."""
    import torch.nn.functional as F
    from segmi.distributed import DistributedModel, global_batch_mean
    g = torch.Generator().manual_seed(17)
    sizes = (3, 2)                                           # ragged shards too
    X = torch.randn(sum(sizes), 3, 8, 8, generator=g)
    T = torch.randint(0, 5, (sum(sizes), 8, 8), generator=g)
    T[0, :6] = 255                                           # rank 0's shard: 3/4 of its first image ignored
    T[4, :1] = 255                                           # rank 1's shard: one row
    off = sum(sizes[:rank])
    xs, ts = X[off:off + sizes[rank]], T[off:off + sizes[rank]]
    torch.manual_seed(3)
    net = torch.nn.Conv2d(3, 5, 3, padding=1)
    dm = DistributedModel(net)
    out = {}
    for mode in ("global", "naive"):
        dm.zero_grad()
        logits = dm(xs)
        lsum = F.cross_entropy(logits, ts, ignore_index=255, reduction="sum")
        cnt = (ts != 255).sum().float()
        if mode == "global":
            loss, den = global_batch_mean(lsum, cnt)
        else:
            loss = lsum / cnt                                # what round 1 did: the mean of per-rank means
        loss.backward()
        dm.finish_gradients()
        lavg = loss.detach().clone()
        dist.all_reduce(lavg)
        out[mode] = (lavg.item() / world, {k: p.grad.clone() for k, p in net.named_parameters()})
    torch.manual_seed(3)
    ref = torch.nn.Conv2d(3, 5, 3, padding=1)
    rl = F.cross_entropy(ref(X), T, ignore_index=255)
    rl.backward()
    err = {m: max((out[m][1][k] - p.grad).abs().max().item() for k, p in ref.named_parameters()) for m in out}
    return {"loss": {m: abs(out[m][0] - rl.item()) for m in out}, "err": err, "den": float(den), "valid": float((T != 255).sum())}


def test_global_batch_loss_semantics_with_unequal_ignore_regions():
    for r in _spawn(_loss_worker):
        assert r["den"] == r["valid"] / 2                     # global valid count / world
        assert r["loss"]["global"] < 1e-6 and r["err"]["global"] < 1e-6, r
        assert r["err"]["naive"] > 1e-4, r                    # the per-rank mean is NOT the reference's semantics here


def _welford_partial(x):
    """[3*C] packed partial exactly as segmi_bn_stats emits it: count (replicated), mean, M2 per channel."""
    C = x.shape[1]
    flat = x.permute(1, 0, 2, 3).reshape(C, -1).double()
    n = flat.shape[1]
    mean = flat.mean(1)
    m2 = ((flat - mean[:, None]) ** 2).sum(1)
    return torch.cat([torch.full((C,), float(n), dtype=torch.float64), mean, m2]).float()


def _chan_merge(parts, nparts, C):
    """Host restatement of segmi_bn_finalize's merge (Chan et al.)."""
    p = parts.view(nparts, 3, C).double()
    n, mean, m2 = p[0, 0].clone(), p[0, 1].clone(), p[0, 2].clone()
    for i in range(1, nparts):
        nb, mb, m2b = p[i, 0], p[i, 1], p[i, 2]
        tot = n + nb
        delta = mb - mean
        mean = mean + delta * nb / tot
        m2 = m2 + m2b + delta * delta * n * nb / tot
        n = tot
    return n, mean, m2


def _syncbn_worker(rank, world):
    from segmi.distributed import SyncBNContext
    g = torch.Generator().manual_seed(9)
    rows = (3, 5)                                   # ragged shards: ranks hold different batch sizes
    X = torch.randn(sum(rows), 6, 4, 4, generator=g) * 3 + 1
    off = sum(rows[:rank])
    x = X[off:off + rows[rank]]
    ctx = SyncBNContext()
    parts, nparts = ctx.gather_stats(_welford_partial(x))
    n, mean, m2 = _chan_merge(parts, nparts, 6)
    # the global count is never exchanged on its own: it is the sum of the counts the partials carry (what segmi_bn_finalize
    # sums on the device), so ragged shards cannot leave a rank with a stale cached value
    count = float(parts.view(nparts, 3, 6)[:, 0, 0].sum())
    ref_mean = X.double().mean((0, 2, 3))
    ref_var = X.double().var((0, 2, 3), unbiased=False)
    sums = torch.arange(12.0) * (rank + 1)
    gs = ctx.reduce_sums(sums)
    return {"nparts": nparts, "count": count, "collectives": ctx.collectives, "n": float(n[0]),
            "mean_err": (mean - ref_mean).abs().max().item(), "var_err": (m2 / n - ref_var).abs().max().item(),
            "sums_ok": torch.equal(gs, torch.arange(12.0) * 3), "local_untouched": torch.equal(sums, torch.arange(12.0) * (rank + 1))}


def test_syncbn_statistic_exchange_matches_global_batch():
    for r in _spawn(_syncbn_worker):
        assert r["nparts"] == 2 and r["count"] == r["n"] == 8 * 16
        assert r["collectives"] == 2          # one all-gather forward, one all-reduce backward — no count exchange
        assert r["mean_err"] < 1e-6 and r["var_err"] < 1e-5
        assert r["sums_ok"] and r["local_untouched"]


def test_single_process_paths_need_no_process_group():
    from segmi.distributed import GradAllReducer, SyncBNContext
    lin = torch.nn.Linear(3, 2)
    red = GradAllReducer(lin.parameters())
    red.zero_grad()
    lin(torch.ones(4, 3)).sum().backward()
    red.finish()
    assert torch.allclose(lin.weight.grad, torch.full((2, 3), 4.0))
    ctx = SyncBNContext()
    p = torch.ones(6)
    assert ctx.gather_stats(p) == (p, 1) and ctx.reduce_sums(p) is p


def test_syncbn_plugin_surface():
    """convert_model / DataParallelWithCallback keep the reference's names and share parameters."""
    import models
    from segmi import nn as snn
    from utils.sync_batchnorm import DataParallelWithCallback, SynchronizedBatchNorm2d, convert_model, patch_replication_callback
    m = models.PSPNet(3, backbone="resnet50", pretrained=False)
    keys = list(m.state_dict().keys())
    w_before = m.layer1[0].bn1.weight
    m2 = convert_model(m)
    bns = [b for b in m2.modules() if isinstance(b, torch.nn.BatchNorm2d)]
    assert bns and all(isinstance(b, SynchronizedBatchNorm2d) and isinstance(b, snn.BatchNorm2d) for b in bns)
    assert list(m2.state_dict().keys()) == keys and m2.layer1[0].bn1.weight is w_before
    dp = DataParallelWithCallback(m2, device_ids=[0])
    assert dp.module is m2 and patch_replication_callback(dp) is dp
    assert all(p.grad is not None for p in m2.parameters())   # bucket views attached


def _sync_many_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from segmi.distributed import SyncBNContext
        g = torch.Generator().manual_seed(100 + rank)
        parts = [torch.randn(3 * c, generator=g) for c in (8, 16, 4)]
        sums = [torch.randn(2 * c, generator=g) for c in (8, 16, 4)]
        one, many = SyncBNContext(), SyncBNContext()
        ref_g = [one.gather_stats(p) for p in parts]
        ref_s = [one.reduce_sums(s) for s in sums]
        got_g = many.gather_stats_many(parts)
        got_s = many.reduce_sums_many(sums)
        ok = all(torch.equal(a[0], b[0]) and a[1] == b[1] == world for a, b in zip(ref_g, got_g))
        ok = ok and all(torch.equal(a, b) for a, b in zip(ref_s, got_s))
        ok = ok and all(s.data_ptr() != gs.data_ptr() for s, gs in zip(sums, got_s))      # local sums stay the parameter gradients
        ret[rank] = (ok, one.collectives, many.collectives)
    finally:
        dist.destroy_process_group()


def test_syncbn_collectives_of_parallel_layers_are_batched():
    """SyncBNContext.gather_stats_many / reduce_sums_many: one all-gather / one all-reduce for several layers' partials, values
    identical to the per-layer calls (world 2, gloo)."""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_sync_many_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    for r in range(world):
        ok, n_one, n_many = ret[r]
        assert ok and n_one == 6 and n_many == 2, ret[r]


def test_bucket_schedule_follows_backward_order():
    """segmi.distributed.bucket_schedule: every bucket holds half of the bytes still to come, clamped to [4, 64] MiB — large
    collectives while the rest of backward can hide them, a small one for the gradients produced last."""
    from segmi.distributed import GradAllReducer, bucket_schedule
    mb = 1 << 20
    caps = bucket_schedule(206 * mb)
    assert sum(caps) == 206 * mb and caps[0] == caps[1] == 64 * mb and caps[-1] <= 6 * mb and all(a >= b for a, b in zip(caps, caps[1:]))
    assert bucket_schedule(3 * mb) == [3 * mb] and sum(bucket_schedule(26 * mb)) == 26 * mb and len(bucket_schedule(26 * mb)) == 4
    assert all(c >= 2 * mb for c in bucket_schedule(1000 * mb)) and max(bucket_schedule(1000 * mb)) == 64 * mb
    # the reducer lays its buckets out accordingly (reverse parameter order = the order backward produces gradients)
    net = torch.nn.Sequential(*[torch.nn.Linear(1024, 1024, bias=False) for _ in range(24)])          # 24 x 4 MiB
    red = GradAllReducer(net.parameters())
    sizes = [b["buf"].numel() * 4 // mb for b in red.buckets]
    assert sum(sizes) == 96 and sizes[0] == 48 and sizes[-1] == 4 and all(a >= b for a, b in zip(sizes, sizes[1:])), sizes
    assert red.buckets[0]["params"][0] is list(net.parameters())[-1]                                    # last layer's gradient arrives first
    red.remove()
    uni = GradAllReducer(net.parameters(), bucket_bytes=16 * mb)
    assert [b["buf"].numel() * 4 // mb for b in uni.buckets] == [16] * 6
    uni.remove()


def test_bucket_layout_with_an_oversized_tensor_keeps_large_buckets():
    """Round 5 regression: PSPNet-R50's gradients contain one 72 MiB filter (the PSP bottleneck).  It closes the bucket in front of it
    early and travels alone; the capacities of the FOLLOWING buckets must come from the bytes actually still to come (half of the
    rest, clamped to [4, 64] MiB) — indexing a precomputed schedule by bucket number cut the last 40 MiB into ten 4 MiB buckets
    (17 collectives + 17 optimizer launches per step instead of 8).  Also: a bucket of a few small vectors does not travel alone in
    front of an oversized tensor."""
    from segmi.distributed import GradAllReducer
    mb = 1 << 20

    def lin(n_mb):
        return torch.nn.Linear(1024, n_mb * 256, bias=False)          # n_mb MiB of fp32

    # registration order = forward order; backward produces the gradients in reverse: 18 MiB head, the 72 MiB filter, then 116 MiB of
    # 4 MiB tensors (the backbone), a tiny BN-like vector right behind a 9 MiB tensor at the very end
    tiny = torch.nn.BatchNorm1d(64)
    net = torch.nn.Sequential(lin(9), tiny, *[lin(4) for _ in range(29)], lin(72), *[lin(6) for _ in range(3)])
    red = GradAllReducer(net.parameters())
    sizes = [b["buf"].numel() * 4 / mb for b in red.buckets]
    assert abs(sum(sizes) - (9 + 116 + 72 + 18)) < 0.01
    assert sizes[0] == 18 and sizes[1] == 72, sizes                   # the oversized tensor alone, the head before it
    assert len(sizes) <= 9 and min(sizes) >= 1.0, sizes               # no run of 4 MiB buckets, no sliver bucket
    assert sizes[2] >= 40 and all(a >= b * 0.99 for a, b in zip(sizes[2:], sizes[3:-1])), sizes   # halves of what remains, in backward order
    red.remove()


def _buffers_worker(rank, world):
    from segmi.distributed import DistributedModel
    torch.manual_seed(7)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 1), torch.nn.BatchNorm2d(4), torch.nn.Conv2d(4, 2, 1), torch.nn.BatchNorm2d(2))
    dm = DistributedModel(net)
    g = torch.Generator().manual_seed(50 + rank)           # every rank its own shard -> its own batch statistics
    net.train()
    dm(torch.randn(4, 3, 5, 5, generator=g) * (1 + rank) + rank)
    before = torch.cat([b.reshape(-1).float() for b in net.buffers()]).clone()
    dm.broadcast_buffers(src=0)
    after = torch.cat([b.reshape(-1).float() for b in net.buffers()])
    return {"before": before, "after": after, "nbt": [int(m.num_batches_tracked) for m in net if isinstance(m, torch.nn.BatchNorm2d)]}


def test_broadcast_buffers_gives_every_rank_rank0s_running_statistics():
    """Without SyncBN the ranks' BatchNorm running statistics follow their own shards; before a validation pass every rank takes
    rank 0's (DistributedModel.broadcast_buffers, one collective) — the one set the reference's single-process DataParallel keeps
    (base/base_trainer.py:33-38).  Integer buffers (num_batches_tracked) are equal anyway and stay untouched."""
    a, b = _spawn(_buffers_worker)
    assert not torch.equal(a["before"], b["before"])                       # they did drift
    assert torch.equal(a["after"], a["before"]) and torch.equal(b["after"], a["before"])
    assert a["nbt"] == b["nbt"] == [1, 1]
