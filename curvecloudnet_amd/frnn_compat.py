"""``frnn``-signature entry points over ``libccn_hip.so``: what the reference binds when it does ``import frnn``.

The reference calls the third-party FRNN package (https://github.com/lxxue/FRNN, un-vendored submodule,
``/root/reference/.gitmodules:4-6``) at exactly two sites:

* ``src/models/utils/point_ops.py:459``  ``dists, idxs, nn, grid = frnn.frnn_grid_points(points1, points2, lengths1, lengths2, K, r)``
* ``src/models/modules/dgcnn.py:172``    ``feats = frnn.frnn_gather(x, idxs, lengths_p2)``

With ``curvecloudnet_amd.frnn_compat.install()`` (``sys.modules["frnn"] = this module``) the reference's unedited files bind
the HIP hash grid (``csrc/ccn_frnn.hip``) through those names; nothing else of the package's surface is used by the
reference, nothing else is offered.  Semantics follow the published ones restated in SURVEY.md App. C (and in
``oracle/frnn_bruteforce.c``, which the GPU test holds this module to): the <= K nearest ``points2`` of every query with
squared distance < r^2, ascending by (distance, index); ``idxs`` int64 and ``dists`` (SQUARED distances) float32, both padded
with -1, rows past ``lengths1`` all -1.  The search is exact -- the grid only prunes -- so ``radius_cell_ratio`` cannot change
the result and ``return_sorted=False`` (any order allowed upstream) returns the sorted lists.  GPU tensors only, like the
package (``TypeError`` from the reference's own wrapper otherwise); no CPU path.
"""
import sys
from collections import namedtuple

import torch

from . import ops
from ._lib import call, lib, ptr, require_gpu, workspace

# what the fourth return value carries, so that a caller can hand it back (``grid=``) for more queries against the same
# points2 / radius -- the reuse upstream's ``_GRID`` tuple allows.  Opaque to the caller.
Grid = namedtuple("Grid", ["table", "points2", "lengths2", "r", "key"])


def _radius(r, b, device):
    if isinstance(r, (float, int)):
        r = torch.full((b,), float(r), dtype=torch.float32)
    r = r.to(torch.float32).reshape(-1)
    if r.numel() == 1:
        r = r.expand(b)
    if r.numel() != b:
        raise ValueError("r must hold one radius or one per cloud")
    return r.contiguous().to(device)


def frnn_grid_points(points1, points2, lengths1=None, lengths2=None, K=-1, r=-1, grid=None, return_nn=False,
                     return_sorted=True, radius_cell_ratio=2.0, **_ignored):
    """-> ``(dists, idxs, nn, grid)`` as ``frnn.frnn_grid_points`` (reference point_ops.py:459).  ``dists``: squared
    distances (B, P1, K) float32; ``idxs`` (B, P1, K) int64; ``nn`` (B, P1, K, 3) the neighbours' coordinates when
    ``return_nn`` else None; ``grid``: pass it back as ``grid=`` to query the same ``points2`` / ``r`` again without a rebuild."""
    if points1.shape[0] != points2.shape[0]:
        raise ValueError("points1 and points2 must have the same batch  dimension")
    if points1.shape[2] != 3 or points2.shape[2] != 3:
        raise ValueError("only 3-D points are supported")
    if K <= 0:
        raise ValueError("K must be positive")
    require_gpu(points1, points2)
    p1 = points1.detach().to(torch.float32).contiguous()
    p2 = points2.detach().to(torch.float32).contiguous()
    b, n1, n2, dev = p1.size(0), p1.size(1), p2.size(1), p1.device
    if lengths1 is None:
        lengths1 = torch.full((b,), n1, dtype=torch.int64, device=dev)
    if lengths2 is None:
        lengths2 = torch.full((b,), n2, dtype=torch.int64, device=dev)
    l1 = lengths1.to(device=dev, dtype=torch.int64).contiguous()
    l2 = lengths2.to(device=dev, dtype=torch.int64).contiguous()
    rr = _radius(r, b, dev)
    key = (p2.data_ptr(), p2._version, n2, b)
    if grid is not None and isinstance(grid, Grid) and grid.key == key and torch.equal(grid.r, rr) and torch.equal(grid.lengths2, l2):
        table = grid.table
    else:
        table = workspace(lib().ccn_frnn_grid_bytes(b, n2), dev)
        call("frnn_grid_build", ptr(p2), ptr(l2), ptr(rr), b, n2, ptr(table), table.numel())
        grid = Grid(table, p2, l2, rr, key)
    idxs = torch.empty((b, n1, K), dtype=torch.int64, device=dev)
    dists = torch.empty((b, n1, K), dtype=torch.float32, device=dev)
    call("frnn_query", ptr(p1), ptr(l1), ptr(rr), b, n1, K, ptr(table), n2, ptr(idxs), ptr(dists), None)
    nn = frnn_gather(p2, idxs, l2) if return_nn else None
    return dists, idxs, nn, grid


class _Gather(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, idxs):
        b, p2, c = x.shape
        _, p1, k = idxs.shape
        dev = x.device
        xm = x.contiguous()
        cloud_ptr = torch.arange(b + 1, dtype=torch.int64, device=dev) * p2     # the padded layout as b "clouds" of p2 rows
        out = torch.empty((b, p1, k, c), dtype=torch.float32, device=dev)
        call("gather_edge_fwd", ptr(xm), c, ptr(idxs), ptr(cloud_ptr), b, p1, k, c, ptr(out), c)
        ctx.save_for_backward(idxs, cloud_ptr)
        ctx.shape = (b, p2, c)
        return out

    @staticmethod
    def backward(ctx, g):
        idxs, cloud_ptr = ctx.saved_tensors
        b, p2, c = ctx.shape
        _, p1, k = idxs.shape
        g = g.contiguous()
        dx = torch.zeros((b, p2, c), dtype=torch.float32, device=g.device)
        call("gather_edge_bwd", ptr(g), c, ptr(idxs), ptr(cloud_ptr), b, p1, k, c, ptr(dx), c)
        return dx, None


def frnn_gather(x, idxs, lengths=None):
    """``frnn.frnn_gather(x (B, P2, C), idxs (B, P1, K), lengths2)`` -> (B, P1, K, C) (reference dgcnn.py:172):
    ``out[b, i, k] = x[b, idxs[b, i, k]]``, zero rows where ``idxs < 0``; differentiable in ``x``.  ``lengths`` is accepted
    for the signature: FRNN's indices never point past a cloud's length."""
    if x.dim() != 3 or idxs.dim() != 3 or x.shape[0] != idxs.shape[0]:
        raise ValueError("frnn_gather: x must be (B, P2, C) and idxs (B, P1, K)")
    require_gpu(x, idxs)
    if x.dtype != torch.float32:
        raise TypeError("frnn_gather: float32 features only")
    return _Gather.apply(x, idxs.to(torch.int64).contiguous())


def install():
    """``sys.modules["frnn"] = curvecloudnet_amd.frnn_compat``: the reference's ``import frnn`` then binds this module."""
    sys.modules["frnn"] = sys.modules[__name__]
    return sys.modules[__name__]


__all__ = ["frnn_grid_points", "frnn_gather", "install", "Grid"]
_ = ops          # (imported for its side effect of configuring the library the same way the step modules do)
