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


def reserve_headroom(device=None, fraction=0.25, small_blocks=128, streams=None):
    """Keep ``fraction`` of the bytes reserved so far (``torch.cuda.max_memory_reserved`` after the warm-up steps) as free
    cached blocks in every stream's pool: a ladder from a quarter of that amount down to 4 MB, two blocks per size, plus
    ``small_blocks`` blocks of just under 1 MB for the small-block pool.  Stops quietly at the first allocation the device
    cannot satisfy (several ranks on one card, a smaller device).  Returns the bytes kept per stream."""
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    total = int(fraction * torch.cuda.max_memory_reserved(device))
    sizes, size = [], max(total // 4, 4 << 20)
    while size >= (4 << 20):
        sizes += [size, size]
        size //= 2
    kept = {}
    for st in (streams if streams is not None else streams_in_use(device)):
        held, got = [], 0
        with torch.cuda.stream(st):
            try:
                for nbytes in sizes:
                    held.append(torch.empty(nbytes, dtype=torch.uint8, device=device))
                    got += nbytes
                for _ in range(small_blocks):
                    held.append(torch.empty((1 << 20) - 512, dtype=torch.uint8, device=device))
                    got += (1 << 20) - 512
            except torch.cuda.OutOfMemoryError:
                pass
            del held
        kept[st] = got
    return kept
