"""Dataset-side curve splitters on the device: the per-sweep pass that produces the
``curve_idxs`` the hot path consumes.  Same results as the reference's ``_get_curves`` methods
(src/data/kitti_dataset.py:73-92, src/data/nuscenes_dataset.py:91-118), bit-exact.
"""
import torch

from ._lib import call, lib, ptr, require_gpu, workspace

CURVE_THRESH = 0.08            # SemKITTI.CURVE_THRESH / SemNuScenes.CURVE_THRESH


def split_curves(points, beam_idxs=None, thresh=CURVE_THRESH):
    """(N,) int64 curve ids of a sweep in acquisition order; ``beam_idxs`` (N,) adds a split at every beam change."""
    require_gpu(points, beam_idxs)
    if points.dim() != 2 or points.size(1) < 3:
        raise ValueError("points must be (N, >=3)")
    pos = points[:, :3].to(torch.float32).contiguous()
    n, dev = pos.size(0), pos.device
    if n == 0:
        return torch.zeros(0, dtype=torch.int64, device=dev)
    beam = None if beam_idxs is None else beam_idxs.to(torch.int64).contiguous()
    curve = torch.empty(n, dtype=torch.int64, device=dev)
    count = torch.empty(1, dtype=torch.int64, device=dev)
    ws = workspace(lib().ccn_curve_split_workspace_bytes(n), dev)
    call("curve_split", ptr(pos), ptr(beam), n, float(thresh), ptr(curve), ptr(count), ptr(ws), ws.numel())
    return curve


def get_curves_kitti(points, thresh=CURVE_THRESH):
    """ref kitti_dataset.py:73-92 (one sequential beam).  The reference returns the same ids as float32 and casts
    them with ``.long()`` when it builds the sample (:59); this returns int64 directly."""
    return split_curves(points, None, thresh)


def get_curves_nuscenes(points, beam_idxs, labels, reflectance, thresh=CURVE_THRESH):
    """ref nuscenes_dataset.py:91-118: stable sort by beam id (device radix sort: plumbing), split, and the inverse
    permutation.  Returns (points, curve_idxs, labels, reflectance, inv_reorder) like the reference."""
    order = torch.sort(beam_idxs, stable=True)[1]
    inverse = torch.empty_like(order)
    inverse[order] = torch.arange(points.size(0), device=order.device)
    points, beam_idxs, labels, reflectance = points[order], beam_idxs[order], labels[order], reflectance[order]
    return points, split_curves(points, beam_idxs, thresh), labels, reflectance, inverse
