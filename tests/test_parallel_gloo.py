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


def test_single_process_is_a_noop():
    from curvecloudnet_amd.parallel import GradientAllReduce
    m = torch.nn.Linear(3, 2)
    sync = GradientAllReduce(m)
    sync.zero_grad()
    m(torch.ones(4, 3)).sum().backward()
    sync.finish()
    assert m.weight.grad is not None and float(m.weight.grad.abs().sum()) > 0
    assert sync.num_bytes == (6 + 2) * 4
