"""Caching-allocator headroom for a training loop (what ``bench.py`` does before its timed region, as a library call).

PyTorch's caching allocator keeps one pool of blocks per stream.  This package allocates on two streams per device -- the
feature stream and the geometry stream of ``ModelBase.prepare`` -- and in steady state the host runs further ahead of the
GPU than during the first steps (nothing synchronises inside a step), so a pool can need a little more than its peak so
far: a ``hipMalloc`` in mid-run, synchronous and slow (3-4 per 20 steps, measured in round 2).  ``reserve_headroom`` allocates
and frees a ladder of blocks on every stream in use: they stay cached in that stream's pool and are split on demand.
Call it once after a few warm-up steps.  It changes no result and needs no GPU-side work.
"""
import torch


def streams_in_use(device):
    """The streams this package allocates on for ``device``: torch's current stream, the geometry stream(s) and (when
    CCN_WGRAD_STREAM=1) the weight-gradient stream."""
    from . import ops, steps
    device = torch.device(device)
    extra = [s for s in list(steps._GEOMETRY_STREAMS.values()) + list(ops._WGRAD_STREAMS.values()) if s.device == device]
    return [torch.cuda.current_stream(device)] + extra


LAST_REPORT = {}        # what the last reserve_headroom call did: {"budget", "wanted", "scaled", "cut_short": [streams]}


def reserve_headroom(device=None, fraction=0.25, small_blocks=128, streams=None, margin_bytes=4 << 30):
    """Keep ``fraction`` of the bytes reserved so far (``torch.cuda.max_memory_reserved`` after the warm-up steps) as free
    cached blocks in every stream's pool (the current stream first): a ladder from a quarter of the amount down to 4 MB, two
    blocks per size, plus ``small_blocks`` blocks of just under 1 MB for the small-block pool.  The total over all streams is
    capped by the device's free memory minus ``margin_bytes`` (``torch.cuda.mem_get_info``) -- ladders that do not fit are
    scaled down together, so that the side streams' pools can never take what the main stream's next allocation needs
    (an allocator out-of-memory retry frees every cached block with synchronous hipFree calls: the stall this helper exists to
    avoid).  Returns the bytes kept per stream; ``memory.LAST_REPORT`` says whether a ladder was cut short."""
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    sts = list(streams if streams is not None else streams_in_use(device))
    base = int(fraction * torch.cuda.max_memory_reserved(device))
    wanted = [base for _ in sts]
    free, _ = torch.cuda.mem_get_info(device)
    budget = max(0, int(free) - int(margin_bytes))
    # (two blocks per size from total / 4 down: a ladder holds ~total bytes; the small blocks come on top)
    need = sum(wanted) + len(sts) * small_blocks * (1 << 20)
    scale = min(1.0, budget / need) if need > 0 else 1.0
    kept, cut = {}, []
    for st, total in zip(sts, wanted):
        total = int(total * scale)
        sizes, size = [], max(total // 4, 4 << 20)
        while size >= (4 << 20) and total >= (8 << 20):
            sizes += [size, size]
            size //= 2
        held, got = [], 0
        with torch.cuda.stream(st):
            try:
                for nbytes in sizes:
                    held.append(torch.empty(nbytes, dtype=torch.uint8, device=device))
                    got += nbytes
                for _ in range(int(small_blocks * scale)):
                    held.append(torch.empty((1 << 20) - 512, dtype=torch.uint8, device=device))
                    got += (1 << 20) - 512
            except torch.cuda.OutOfMemoryError:
                cut.append(st)
            del held
        kept[st] = got
    LAST_REPORT.clear()
    LAST_REPORT.update({"budget": budget, "wanted": sum(wanted), "scaled": scale < 1.0, "cut_short": cut})
    return kept
