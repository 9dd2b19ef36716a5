"""Synthetic LiDAR-like curve clouds: the shape of data the hot path consumes.

A cloud is ``Q`` polylines ("curves") packed back to back: ``pos (N,3) f32``, reflectance
``x (N,1) f32``, ``curve_idxs (N,) i64`` (local, non-decreasing from 0).  A batch adds
``batch (N,) i64``.  Deterministic per ``cloud_id`` (``torch.Generator().manual_seed(1234+id)``).
"""
import math
from types import SimpleNamespace

import torch


def make_cloud(cloud_id=0, n_curves=2048, min_len=1, max_len=48, mixed_lengths=False, step=0.0035, lengths=None):
    g = torch.Generator().manual_seed(1234 + cloud_id)
    if lengths is not None:
        lens = torch.as_tensor(lengths, dtype=torch.long)
        n_curves = lens.numel()
    elif mixed_lengths:   # config 5: log-normal(ln 16, 0.9) clamped to [1, 512]
        lens = torch.exp(torch.randn(n_curves, generator=g) * 0.9 + math.log(16.0)).round().clamp(1, 512).long()
    else:
        lens = torch.randint(min_len, max_len + 1, (n_curves,), generator=g)
    lmax = int(lens.max())
    start = torch.rand(n_curves, 3, generator=g) * torch.tensor([6.0, 6.0, 0.6]) - torch.tensor([3.0, 3.0, 0.3])
    heading = torch.rand(n_curves, 1, generator=g) * (2 * math.pi) + \
        torch.cumsum(torch.randn(n_curves, lmax, generator=g) * 0.05, dim=1)
    hop = step * (0.8 + 0.4 * torch.rand(n_curves, lmax, generator=g))
    delta = torch.stack([hop * torch.cos(heading), hop * torch.sin(heading), torch.zeros_like(hop)], dim=-1)
    delta[:, 0] = 0
    pts = start[:, None, :] + torch.cumsum(delta, dim=1)
    live = torch.arange(lmax)[None, :] < lens[:, None]
    pos = pts[live].float().contiguous()
    curve = torch.arange(n_curves)[:, None].expand(-1, lmax)[live].contiguous()
    x = torch.rand(pos.size(0), 1, generator=g)
    return SimpleNamespace(x=x, pos=pos, curve_idxs=curve, lengths=lens)


def make_batch(cloud_ids, **kw):
    clouds = [make_cloud(c, **kw) for c in cloud_ids]
    return SimpleNamespace(
        x=torch.cat([c.x for c in clouds]),
        pos=torch.cat([c.pos for c in clouds]),
        curve_idxs=torch.cat([c.curve_idxs for c in clouds]),
        batch=torch.cat([torch.full((c.pos.size(0),), i, dtype=torch.long) for i, c in enumerate(clouds)]),
        num_clouds=len(clouds),
    )


def to_device(data, device):
    out = SimpleNamespace(**vars(data))
    for k, v in vars(out).items():
        if torch.is_tensor(v):
            setattr(out, k, v.to(device))
    return out
