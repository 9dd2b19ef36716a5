"""CPU: the N>1 path -- cloud sharding + bucketed gradient all-reduce, world_size 2 over gloo."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from curvecloudnet_amd.parallel import GradientAllReduce, init_process_group_from_env, shard_clouds
    r, w, _ = init_process_group_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.ReLU(), torch.nn.Linear(16, 16), torch.nn.ReLU(),
                                torch.nn.Linear(16, 3))
    sync = GradientAllReduce(model, bucket_bytes=600)          # several small buckets
    assert len(sync.buckets) > 1
    clouds = shard_clouds(range(4), rank, world)
    assert clouds == [rank, rank + 2]
    for step in range(2):
        sync.zero_grad()
        losses = []
        for c in clouds:
            g = torch.Generator().manual_seed(100 + c)
            x, y = torch.randn(10, 6, generator=g), torch.randn(10, 3, generator=g)
            losses.append(((model(x) - y) ** 2).mean() / len(clouds))
        if step == 0:
            sum(losses).backward()                  # one backward per step: all-reduce overlaps it
        else:
            with sync.no_sync():                    # accumulation over several backward passes
                for l in losses:
                    l.backward()
        sync.finish()
        with torch.no_grad():
            for p in model.parameters():
                p -= 0.1 * p.grad
    flat = torch.cat([p.detach().flatten() for p in model.parameters()])
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    if rank == 0:
        out.put([g.tolist() for g in gathered])
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_allreduce_world2_matches_single_process():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    got = out.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    a, b = torch.tensor(got[0]), torch.tensor(got[1])
    assert torch.equal(a, b)                                    # replicas stay identical
    # single-process run over all 4 clouds = the average of the two ranks' gradients
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.ReLU(), torch.nn.Linear(16, 16), torch.nn.ReLU(),
                                torch.nn.Linear(16, 3))
    for step in range(2):
        model.zero_grad()
        for c in range(4):
            g = torch.Generator().manual_seed(100 + c)
            x, y = torch.randn(10, 6, generator=g), torch.randn(10, 3, generator=g)
            (((model(x) - y) ** 2).mean() / 4).backward()
        with torch.no_grad():
            for p in model.parameters():
                p -= 0.1 * p.grad
    want = torch.cat([p.detach().flatten() for p in model.parameters()])
    assert float((a - want).abs().max()) < 1e-6


def _worker_uneven(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from curvecloudnet_amd.parallel import GradientAllReduce, init_process_group_from_env, shard_clouds
    init_process_group_from_env(backend="gloo")
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.ReLU(), torch.nn.Linear(16, 3))
    sync = GradientAllReduce(model, bucket_bytes=300)
    clouds = shard_clouds(range(6), rank, world)              # 6 clouds over 4 ranks: two ranks take 2, two take 1
    sync.zero_grad()
    points = 0
    loss = 0
    for c in clouds:
        g = torch.Generator().manual_seed(200 + c)
        n = 5 + 7 * c                                         # clouds of very different sizes
        x, y = torch.randn(n, 6, generator=g), torch.randn(n, 3, generator=g)
        points += n
        loss = loss + ((model(x) - y) ** 2).mean() / len(clouds)
    loss.backward()
    sync.finish()
    flat = torch.cat([p.grad.flatten() for p in model.parameters()])
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    counts = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([points]))
    if rank == 0:
        out.put(([t.tolist() for t in gathered], [int(c) for c in counts], dict(sync.stats)))
    dist.barrier()
    dist.destroy_process_group()


def test_world4_uneven_clouds_per_rank():
    """Four gloo ranks, six clouds of very different sizes: shard_clouds gives the ranks 2 / 2 / 1 / 1 whole clouds, the
    load-imbalance figure of a run (max / mean points per rank, SURVEY section 8e "Caveat") is far from 1, the replicas stay
    identical, and the reduced gradient is the mean over the ranks of each rank's own mean-over-its-clouds gradient --
    what one process per GPU with per-rank batches computes (no cross-rank weighting by points)."""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    world = 4
    procs = [ctx.Process(target=_worker_uneven, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    grads, points, stats = out.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    grads = [torch.tensor(g) for g in grads]
    for g in grads[1:]:
        assert torch.equal(g, grads[0])
    assert points == [5 + 7 * 0 + 5 + 7 * 4, 5 + 7 * 1 + 5 + 7 * 5, 5 + 7 * 2, 5 + 7 * 3]
    imbalance = max(points) / (sum(points) / len(points))
    assert imbalance > 1.5
    assert stats["bytes_reduced"] > 0 and stats["steps"] == 1
    want = 0
    for rank in range(world):
        torch.manual_seed(0)
        m = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.ReLU(), torch.nn.Linear(16, 3))
        clouds = list(range(6))[rank::world]
        for c in clouds:
            g = torch.Generator().manual_seed(200 + c)
            n = 5 + 7 * c
            x, y = torch.randn(n, 6, generator=g), torch.randn(n, 3, generator=g)
            (((m(x) - y) ** 2).mean() / len(clouds)).backward()
        want = want + torch.cat([p.grad.flatten() for p in m.parameters()]) / world
    assert float((grads[0] - want).abs().max()) < 1e-6


class _FusedLinear(torch.autograd.Function):
    """CPU stand-in for the HIP layers' main-grad protocol (ops.LinearBNAct): the weight gradient is ADDED into the
    bucket view by the layer itself, autograd gets None for it, and the all-reduce is told through note_use / use_done."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x)
        ctx.w = w
        w._ccn_sync.note_use(w)
        return x @ w.detach().t()

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        w = ctx.w
        w._ccn_main_grad.add_(g.t() @ x)
        w._ccn_sync.use_done(w)
        return g @ w.detach(), None


class _TwoUse(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.a = torch.nn.Parameter(torch.randn(5, 7))       # odd sizes: slots must still start 16-byte aligned
        self.b = torch.nn.Parameter(torch.randn(3, 5))
        self.c = torch.nn.Parameter(torch.randn(3))

    def forward(self, x):
        return _FusedLinear.apply(_FusedLinear.apply(x, self.a).relu(), self.b) + self.c


def _worker_two_uses(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from curvecloudnet_amd.parallel import GradientAllReduce, init_process_group_from_env
    init_process_group_from_env(backend="gloo")
    torch.manual_seed(0)
    model = _TwoUse()
    sync = GradientAllReduce(model, bucket_bytes=64)            # one bucket per parameter or two
    for flat, plist, offs in sync.buckets:
        assert all(o % 4 == 0 for o in offs) and flat.numel() % 4 == 0
        for p, o in zip(plist, offs):
            assert p.grad.data_ptr() == flat[o:].data_ptr()
    g = torch.Generator().manual_seed(10 + rank)
    x1, x2 = torch.randn(6, 7, generator=g), torch.randn(4, 7, generator=g)
    sync.zero_grad()
    # the model runs TWICE before one backward pass: every fused parameter has two outstanding uses, and its bucket must
    # be reduced after the second product, exactly once
    (model(x1).square().mean() + model(x2).square().mean()).backward()
    sync.finish()
    assert sync.reduce_calls == len(sync.buckets), (sync.reduce_calls, len(sync.buckets))
    flat = torch.cat([p.grad.flatten() for p in model.parameters()])
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    if rank == 0:
        out.put([t.tolist() for t in gathered])
    dist.barrier()
    dist.destroy_process_group()


def test_fused_gradients_with_two_uses_per_backward_world2():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_two_uses, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    got = out.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    a, b = torch.tensor(got[0]), torch.tensor(got[1])
    assert torch.equal(a, b)
    # reference: plain autograd on both ranks' inputs, averaged
    want = 0
    for rank in range(2):
        torch.manual_seed(0)
        m = _TwoUse()
        g = torch.Generator().manual_seed(10 + rank)
        x1, x2 = torch.randn(6, 7, generator=g), torch.randn(4, 7, generator=g)

        def f(x):
            return (x @ m.a.t()).relu() @ m.b.t() + m.c
        (f(x1).square().mean() + f(x2).square().mean()).backward()
        want = want + torch.cat([p.grad.flatten() for p in m.parameters()]) / 2
    assert float((a - want).abs().max()) < 1e-6


def test_late_gradient_after_reduce_raises():
    """Accumulating over two backward passes WITHOUT no_sync() would leave the second pass un-reduced: it must raise."""
    import pytest
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), CCN_SINGLE_RANK_GROUP="gloo")
    from curvecloudnet_amd.parallel import GradientAllReduce
    try:
        dist.init_process_group(backend="gloo", rank=0, world_size=1)
        m = torch.nn.Linear(3, 2)
        sync = GradientAllReduce(m)
        assert sync.reduces
        sync.zero_grad()
        m(torch.ones(4, 3)).sum().backward()
        with pytest.raises(RuntimeError, match="no_sync"):
            m(torch.ones(4, 3)).sum().backward()
        sync.finish()
        # the sanctioned form
        sync.zero_grad()
        with sync.no_sync():
            m(torch.ones(4, 3)).sum().backward()
        m(torch.ones(4, 3)).sum().backward()
        sync.finish()
        assert float(m.bias.grad[0]) == 8.0
    finally:
        os.environ.pop("CCN_SINGLE_RANK_GROUP", None)
        if dist.is_initialized():
            dist.destroy_process_group()


def test_bucket_slots_are_16_byte_aligned_for_the_shipped_configs():
    """with use_bias=True the 55- / 50-class output bias used to push every weight behind it off 16-byte
    alignment (a2d2: 33 of 61 Linear weights)."""
    from curvecloudnet_amd import configs
    from curvecloudnet_amd.model import build_model
    from curvecloudnet_amd.parallel import GradientAllReduce
    for cfg, in_dim, n_out in ((configs.a2d2_config(0.125), 4, 55), (configs.shapenet_seg_config(0.125), 3, 50)):
        model = build_model(cfg, in_dim, n_out)
        assert any(p.numel() % 4 for p in model.parameters())
        sync = GradientAllReduce(model, bucket_bytes=1 << 16)
        for flat, plist, offs in sync.buckets:
            for p, o in zip(plist, offs):
                assert o % 4 == 0 and (p.grad.data_ptr() - flat.data_ptr()) % 16 == 0


def test_single_process_is_a_noop():
    from curvecloudnet_amd.parallel import GradientAllReduce
    m = torch.nn.Linear(3, 2)
    sync = GradientAllReduce(m)
    sync.zero_grad()
    m(torch.ones(4, 3)).sum().backward()
    sync.finish()
    assert m.weight.grad is not None and float(m.weight.grad.abs().sum()) > 0
    assert sync.num_bytes == (8 + 4) * 4          # slots of 6 and 2 floats rounded up to 16-byte multiples
