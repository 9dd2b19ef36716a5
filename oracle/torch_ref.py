"""CPU oracle: a plain PyTorch-CPU / numpy restatement of the CurveCloudNet hot path.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  Every function cites the reference
``file:line`` (relative to the upstream tree) whose behaviour it restates.  The integer /
index functions are pinned bit-for-bit against the reference's own code by
``oracle/gen_golden.py`` (golden vectors in ``tests/golden``, checked by
``tests/test_oracle_golden.py``); floating point outputs are pinned to 1e-5 there.

Parity status: curve functions (A1-A9 of SURVEY.md section 8a) are PINNED by golden vectors made
from the imported reference.  FRNN (``frnn_grid_points``), PyG ``MLP``/``softmax`` and
torch_scatter are third-party packages that are NOT vendored in the reference tree (empty
submodule, version pin unrecoverable): for those this file restates the published semantics
 and parity is UNPINNED -- anchored only on the reference's call sites.

Arithmetic conventions that the HIP kernels reproduce exactly (measured on torch 2.10 CPU):
  * ``norm3(d) = sqrt(fma(dz,dz, fma(dy,dy, dx*dx)))``  (what ``torch.linalg.norm`` does on CPU)
  * ``cumsum`` over float32 accumulates in float64 and rounds every output to float32
  * ``index_add_`` / scatter-add over float32 adds sequentially in index order
  * FRNN squared distance ``d2 = fma(dz,dz, fma(dy,dy, dx*dx))``, neighbour iff ``d2 < r*r``,
    order (d2, index) ascending  (tie order is unspecified upstream)
"""
import copy
import ctypes
import math
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

_HERE = os.path.dirname(os.path.abspath(__file__))

# --------------------------------------------------------------------------------------
# A1 / A2 : segment pointers and curve-id globalisation
# --------------------------------------------------------------------------------------


def segment_starts(ids, with_ends=False):
    """Start offset of every run of a sorted id vector (ref point_ops.py:47-54 ``batch2ptr``).

    Interior starts only (``R-1`` entries) unless ``with_ends`` (then ``[0, ..., N]``)."""
    ids = ids.long()
    step = ids[1:] - ids[:-1]
    if bool((step < 0).any()):
        raise AssertionError("ids must be sorted")
    starts = torch.nonzero(step > 0).flatten() + 1
    if with_ends:
        n = torch.tensor([ids.numel()], dtype=starts.dtype)
        starts = torch.cat([torch.zeros(1, dtype=starts.dtype), starts, n])
    return starts


def curve_ids_global(point2curveidx, batch):
    """Per-cloud curve ids -> batch-global ids (ref point_ops.py:20-44).

    A single-cloud batch returns the input object untouched (quirk Q8)."""
    bounds = segment_starts(batch, with_ends=True)
    n_clouds = bounds.numel() - 1
    if n_clouds == 1:
        return point2curveidx
    last_pt = bounds[1:-1] - 1                       # last point of clouds 0..B-2
    curves_in_cloud = point2curveidx[last_pt] + 1
    offs = torch.cat([torch.zeros(1, dtype=curves_in_cloud.dtype), torch.cumsum(curves_in_cloud, 0)])
    return point2curveidx + offs[batch]


def curve_start_of_point(glob):
    """Index of the first point of the curve each point lies on (ref fps_ops.py:24-25)."""
    first = torch.cat([torch.zeros(1, dtype=torch.long), segment_starts(glob)])
    return first[glob]


# --------------------------------------------------------------------------------------
# A3 : per-point absolute feature differences along the curve
# --------------------------------------------------------------------------------------


def feature_diffs(x, point2curveidx, batch):
    """|mean of the in-curve forward/backward finite differences| (ref fast_conv1d.py:190-205)."""
    g = curve_ids_global(point2curveidx, batch)
    linked = (g[1:] == g[:-1])
    step = torch.where(linked[:, None], x[1:] - x[:-1], torch.zeros((), dtype=x.dtype))
    zrow = torch.zeros((1, x.size(1)), dtype=x.dtype)
    zflag = torch.zeros(1, dtype=x.dtype)
    ahead = torch.cat([step, zrow], 0)               # edge i -> i+1
    behind = torch.cat([zrow, step], 0)              # edge i-1 -> i
    n_edges = torch.cat([linked.to(x.dtype), zflag]) + torch.cat([zflag, linked.to(x.dtype)])
    return routed_abs((ahead + behind) / torch.clamp(n_edges, min=1)[:, None])


# --------------------------------------------------------------------------------------
# A4-A6 : symmetric curve convolution
# --------------------------------------------------------------------------------------


# --------------------------------------------------------------------------------------
# bf16 MLP mode (BASELINE configs 3 / 5): emulation of "operands rounded to bf16, fp32 accumulation" for the
# forward, data-gradient and weight-gradient products.
# --------------------------------------------------------------------------------------
MLP_DTYPE = "fp32"

# Feature arithmetic dtype of an adjudication run (tests only): None = the dtype the tensors arrive in (fp32, the reference's
# arithmetic).  torch.float64 with a ``.double()`` model: every feature computation in fp64 while positions, and with them
# every index decision (sampling, neighbour search, curve groups), stay the reference's fp32 -- "the value the network
# defines on these inputs", against which the fp32 CPU oracle and the GPU are both measured.
FEATURE_DTYPE = None


def _feat(t):
    """A position-derived tensor on its way into the feature path."""
    return t if FEATURE_DTYPE is None else t.to(FEATURE_DTYPE)


def set_mlp_dtype(name):
    """"bf16": operands of the forward / data-gradient / weight-gradient products rounded to bf16.  "fp16" (BASELINE
    configs[4], "fp16 features"): FORWARD operands rounded to fp16, gradient products to bf16 (what the product does:
    gradients leave fp16's normal range without loss scaling)."""
    global MLP_DTYPE
    assert name in ("fp32", "bf16", "fp16")
    MLP_DTYPE = name


def _bf16(t):
    return t.to(torch.bfloat16).to(t.dtype)


# 16-bit STORAGE form of the bf16 mode (the product's ops.STORE16): a hidden MLP activation whose only consumer is the next
# Linear is kept as bf16 rows.  Its forward value is the one both forms round before the product anyway; its GRADIENT is
# stored as bf16 too, i.e. rounded once more than in the fp32-storage form.
STORE16 = True


class _RoundGradBF16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return _bf16(g)


def _fwd16(t):
    return t.to(torch.float16 if MLP_DTYPE == "fp16" else torch.bfloat16).to(t.dtype)


class _LinearBF16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        y = _fwd16(x) @ _fwd16(w).t()
        return y if b is None else y + b

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        # (16-bit storage form of the fp16 mode: the forward operand is KEPT as fp16 rows, the weight-gradient product -- a
        # bf16 product -- takes bf16(fp16(x)))
        xs = x.half().to(x.dtype) if (MLP_DTYPE == "fp16" and STORE16) else x
        return _bf16(g) @ _bf16(w), _bf16(g).t() @ _bf16(xs), (g.sum(0) if ctx.has_bias else None)


class _Conv1dBF16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, pad):
        ctx.save_for_backward(x, w)
        ctx.pad, ctx.has_bias = pad, b is not None
        return F.conv1d(_fwd16(x), _fwd16(w), b, stride=1, padding=pad)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        dx = torch.nn.grad.conv1d_input(x.shape, _bf16(w), _bf16(g), stride=1, padding=ctx.pad)
        xs = x.half().to(x.dtype) if (MLP_DTYPE == "fp16" and STORE16) else x       # (see _LinearBF16.backward)
        dw = torch.nn.grad.conv1d_weight(_bf16(xs), w.shape, _bf16(g), stride=1, padding=ctx.pad)
        return dx, dw, (g.sum(dim=(0, 2)) if ctx.has_bias else None), None


# ---- activation with a routing trace (test hook, same idea as MAX_TRACE below).  ReLU / LeakyReLU are not differentiable
# at 0: where a pre-activation lies within the CPU/GPU forward difference (~1e-5 at depth) of zero, the two sides take
# different slopes and ONE row's contribution to a weight gradient flips -- 1/sqrt(rows) of that gradient entry, far above
# any fp32 tolerance.  With ACT_TRACE["force"] set to the product's per-layer sign tables (in forward order), the oracle
# takes the slope the GPU took; the number of entries where it would have chosen differently and the largest |z| among
# them are recorded.
ACT_TRACE = None        # None | {"force": [bool tables...], "mismatch": 0, "entries": 0, "max_abs": 0.0}


class _RoutedAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, mask, slope):
        ctx.save_for_backward(mask)
        ctx.slope = slope
        return torch.where(mask, z, z * slope)

    @staticmethod
    def backward(ctx, g):
        (mask,) = ctx.saved_tensors
        return torch.where(mask, g, g * ctx.slope), None, None


def routed_abs(v):
    """|v| (ref fast_conv1d.py:205).  Not differentiable at 0 either: with ACT_TRACE set, the sign the GPU computed (an
    int8 table of -1 / 0 / +1, same forward order as the activation tables) is used, out = sign * v."""
    trace = ACT_TRACE
    if trace is None:
        return torch.abs(v)
    sign = trace["force"].pop(0)
    assert sign.shape == v.shape and sign.dtype == torch.int8, ("abs trace out of step", tuple(sign.shape), tuple(v.shape))
    diff = torch.sign(v.detach()).to(torch.int8) != sign
    n = int(diff.sum())
    trace["mismatch"] += n
    trace["entries"] += v.numel()
    trace.setdefault("per_layer", []).append(("abs", tuple(v.shape), n))
    if n:
        trace["max_abs"] = max(trace["max_abs"], float(v.detach()[diff].abs().max()))
    return v * sign.to(v.dtype)


def activation(z, kind):
    """relu / leaky_relu(0.01) (ref base.py:90-125 via PyG MLP; fast_conv1d.py:73,143)."""
    trace = ACT_TRACE
    if trace is None:
        return F.relu(z) if kind == "relu" else F.leaky_relu(z)
    mask = trace["force"].pop(0)
    assert mask.shape == z.shape, ("activation trace out of step", tuple(mask.shape), tuple(z.shape))
    own = z.detach() > 0
    diff = own != mask
    n = int(diff.sum())
    trace["mismatch"] += n
    trace["entries"] += z.numel()
    trace.setdefault("per_layer", []).append((kind, tuple(z.shape), n))       # (forward order: which layer's kinks flipped)
    if n:
        trace["max_abs"] = max(trace["max_abs"], float(z.detach()[diff].abs().max()))
    return _RoutedAct.apply(z, mask, 0.0 if kind == "relu" else 0.01)


def linear(x, lin):
    if MLP_DTYPE in ("bf16", "fp16"):
        return _LinearBF16.apply(x, lin.weight, lin.bias)
    return lin(x)


class SymmetricConv1d(nn.Module):
    """Conv weights store the centre tap and one side; the other side mirrors it
    (ref fast_conv1d.py:148-187).  Parameter names/shapes match the reference's ``_ConvNd``."""

    def __init__(self, in_channels, out_channels, half_taps, bias=True):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, half_taps))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        # same initialisation as torch's _ConvNd.reset_parameters
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in = in_channels * half_taps
            bound = 1 / math.sqrt(fan_in) if fan_in > 0 else 0
            nn.init.uniform_(self.bias, -bound, bound)

    def full_weight(self):
        w = self.weight
        if w.size(2) > 1:
            w = torch.cat([torch.flip(w[:, :, 1:], dims=[2]), w], dim=2)
        return w

    def forward(self, seq):                           # seq: (L, C_in) one zero-padded sequence
        w = self.full_weight()
        if MLP_DTYPE in ("bf16", "fp16"):
            y = _Conv1dBF16.apply(seq.t().unsqueeze(0), w, self.bias, w.size(2) // 2)
        else:
            y = F.conv1d(seq.t().unsqueeze(0), w, self.bias, stride=1, padding=w.size(2) // 2)
        return y.squeeze(0).t()


def _conv_layers(feat_dims, kernel_size, bias, diff_first_only, with_diff):
    convs, norms = [], []
    for i in range(1, len(feat_dims)):
        doubled = with_diff and (i == 1 or not diff_first_only)
        cin = feat_dims[i - 1] * 2 if doubled else feat_dims[i - 1]
        convs.append(SymmetricConv1d(cin, feat_dims[i], kernel_size // 2 + 1, bias=bias))
        norms.append(nn.BatchNorm1d(feat_dims[i]))
    return nn.ModuleList(convs), nn.ModuleList(norms)


def _separator_layout(glob, n_points, pad, with_ends):
    """Row index of every real point inside the zero-separated sequence, and its length
    (ref fast_conv1d.py:48-61 / 115-126)."""
    starts = segment_starts(glob, with_ends=with_ends)
    n_sep = starts.numel()
    n_rows = n_points + n_sep * pad
    is_real = torch.ones(n_rows, dtype=torch.bool)
    if pad > 0 and n_sep > 0:
        sep = (starts[:, None] + torch.arange(n_sep * pad).view(n_sep, pad)).flatten()
        is_real[sep] = False
    return torch.nonzero(is_real).flatten(), n_rows


class SymmetricCurve1DConvFastV1(nn.Module):
    """ref fast_conv1d.py:78-145: per layer diff-concat, separators of k//2 zero rows between
    curves, conv, BN over the real rows only, LeakyReLU."""

    def __init__(self, feat_dims=(64, 64, 128), kernel_size=5, bias=True, with_xyz=False, with_diff=False):
        super().__init__()
        self.kernel_size, self.feat_dims = kernel_size, feat_dims
        self.with_xyz, self.with_diff = with_xyz, with_diff
        self.conv_modules, self.norm_modules = _conv_layers(feat_dims, kernel_size, bias, False, with_diff)

    def forward(self, x, pos, batch, point2curveidx, **kwargs):
        g = curve_ids_global(point2curveidx, batch)
        if self.with_xyz:
            x = _feat(pos) if x is None else torch.cat([x, _feat(pos)], dim=1)
        rows, n_rows = _separator_layout(g, x.size(0), self.kernel_size // 2 if self.kernel_size > 1 else 0, False)
        for conv, norm in zip(self.conv_modules, self.norm_modules):
            if self.with_diff:
                x = torch.cat([x, feature_diffs(x, point2curveidx, batch)], dim=1)
            seq = torch.zeros((n_rows, x.size(1)), dtype=x.dtype).index_copy(0, rows, x)
            x = conv(seq)[rows]
            x = activation(norm(x), "leaky_relu")
        return x, pos, batch, point2curveidx


class SymmetricCurve1DConvV2(nn.Module):
    """ref fast_conv1d.py:11-75: diff once, separators of (k//2)*n_layers rows between curves AND
    at both ends, conv+BN+LeakyReLU on the whole padded sequence (quirk Q2), gather once."""

    def __init__(self, feat_dims=(64, 64, 128), kernel_size=5, bias=True, with_xyz=False, with_diff=False):
        super().__init__()
        self.kernel_size, self.feat_dims = kernel_size, feat_dims
        self.with_xyz, self.with_diff = with_xyz, with_diff
        self.conv_modules, self.norm_modules = _conv_layers(feat_dims, kernel_size, bias, True, with_diff)

    def forward(self, x, pos, batch, point2curveidx, **kwargs):
        g = curve_ids_global(point2curveidx, batch)
        if self.with_xyz:
            x = _feat(pos) if x is None else torch.cat([x, _feat(pos)], dim=1)
        pad = (self.kernel_size // 2) * (len(self.feat_dims) - 1) if self.kernel_size > 1 else 0
        rows, n_rows = _separator_layout(g, x.size(0), pad, True)
        if self.with_diff:
            x = torch.cat([x, feature_diffs(x, point2curveidx, batch)], dim=1)
        seq = torch.zeros((n_rows, x.size(1)), dtype=x.dtype).index_copy(0, rows, x)
        for conv, norm in zip(self.conv_modules, self.norm_modules):
            seq = activation(norm(conv(seq)), "leaky_relu")
        return seq[rows], pos, batch, point2curveidx


# --------------------------------------------------------------------------------------
# A7 : arclength sub-sampling along curves
# --------------------------------------------------------------------------------------


def _edge_lengths(pos, glob):
    step = pos[1:] - pos[:-1]
    length = torch.linalg.norm(step, dim=-1)
    return torch.where(glob[1:] == glob[:-1], length, torch.zeros((), dtype=length.dtype))


def curve_fps(pos, batch, point2curveidx, spacing, u):
    """Keep one point per ``spacing`` of arclength on each curve (ref fps_ops.py:16-39).

    ``u`` is the reference's ``torch.rand(1)`` draw (float32 tensor of shape (1,)), injected so the
    result is reproducible (quirk Q10)."""
    g = curve_ids_global(point2curveidx, batch)
    start = curve_start_of_point(g)
    run = torch.cat([torch.zeros(1, dtype=pos.dtype), torch.cumsum(_edge_lengths(pos, g), dim=0)])
    arclen = run - run[start]
    arclen = arclen + ((start * 117 * u) % spacing)
    bucket = torch.round(arclen / spacing)
    keep = torch.cat([torch.ones(1, dtype=torch.bool), bucket[1:] != bucket[:-1]])
    keep[start] = True
    return torch.nonzero(keep).flatten()


# --------------------------------------------------------------------------------------
# A8 : radius grouping along a curve (queries are a subset of the points)
# --------------------------------------------------------------------------------------


def _alternating_offsets(reach):
    """0, -1, +1, -2, +2, ... , -reach, +reach  (ref point_ops.py:170-171)."""
    mag = torch.arange(1, reach + 1)
    return torch.cat([torch.zeros(1, dtype=torch.long), torch.stack([-mag, mag], dim=1).flatten()])


def curve_hop_budget(pos, glob, radius):
    """Per-curve neighbour budget ceil(radius / mean edge), mean = length / #points, inf -> 1
    (ref point_ops.py:149-162).  Returns (budget float32 (Q,), points-per-curve float32 (Q,))."""
    n_curves = int(glob.max().item()) + 1
    length = torch.zeros(n_curves, dtype=pos.dtype).index_add_(0, glob[1:], _edge_lengths(pos, glob))
    count = torch.zeros(n_curves, dtype=pos.dtype).index_add_(0, glob, torch.ones(glob.numel(), dtype=pos.dtype))
    budget = torch.ceil(radius / (length / count))
    budget[torch.isinf(budget)] = 1
    return budget, count


def curve_radius_group(pos, idx, point2curveidx, batch, radius):
    """Edges (row = query number, col = point index) from every sampled point ``idx[q]`` to its
    nearest points along the same curve (ref point_ops.py:143-193).

    Reproduces quirk Q3: the per-curve budget table is indexed with the *local* curve id."""
    g = curve_ids_global(point2curveidx, batch)
    budget, count = curve_hop_budget(pos, g, radius)
    reach = int(min(budget.max().item(), count.max().item()))
    offs = _alternating_offsets(reach)
    cand = idx[:, None] + offs[None, :]
    inside = (cand >= 0) & (cand < pos.size(0))
    cand = torch.where(inside, cand, torch.zeros((), dtype=cand.dtype))
    ok = inside & (g[cand] == g[idx][:, None])
    allowed = budget[point2curveidx[idx].long()]
    ok = ok & (torch.cumsum(ok, dim=1) <= allowed[:, None])
    q = torch.arange(idx.numel())[:, None].expand_as(cand)
    return q[ok], cand[ok]


# --------------------------------------------------------------------------------------
# A9 : every point -> k nearest sampled points on its own curve, and the interpolation
# --------------------------------------------------------------------------------------


def curve_knn_superset(pos, idx, point2curveidx, batch, k):
    """(row = point index, col = position inside ``idx``) (ref point_ops.py:196-260)."""
    g = curve_ids_global(point2curveidx, batch)
    n, m = pos.size(0), idx.numel()
    taken = torch.zeros(n, dtype=torch.bool)
    taken[idx] = True
    upto = torch.cumsum(taken, dim=0)                 # number of samples at positions <= i
    offs = _alternating_offsets(k + 1)
    cand = upto[:, None] + offs[None, :]
    inside = (cand >= 0) & (cand < m)
    cand = torch.where(inside, cand, torch.zeros((), dtype=cand.dtype))
    cand_pt = idx[cand]
    ok = inside & (g[cand_pt] == g[:, None])
    dist = torch.linalg.norm(pos[cand_pt] - pos[:, None, :], dim=-1)
    dist = torch.where(ok, dist, torch.full((), 100.0, dtype=dist.dtype))
    order = torch.argsort(dist, dim=1)
    cand = torch.gather(cand, 1, order)
    ok = torch.gather(ok, 1, order)
    ok = ok & (torch.cumsum(ok, dim=1) <= k)
    if not bool(ok.any(dim=1).all()):
        raise AssertionError("a point has no sampled neighbour on its curve")
    rows = torch.arange(n)[:, None].expand_as(cand)
    return rows[ok], cand[ok]


def curve_interpolate(x, idx, pos_y, batch_y, point2curveidx_y, k):
    """Inverse-squared-distance interpolation along curves (ref point_ops.py:344-355)."""
    pos_x = pos_y[idx]
    with torch.no_grad():
        y_idx, x_idx = curve_knn_superset(pos_y, idx, point2curveidx_y, batch_y, k)
        d = torch.linalg.norm(pos_x[x_idx] - pos_y[y_idx], dim=-1, keepdim=True) ** 2
        w = (1.0 / torch.clamp(d, min=1e-16)).to(x.dtype)
    n = pos_y.size(0)
    num = torch.zeros((n, x.size(1)), dtype=x.dtype).index_add(0, y_idx, x[x_idx] * w)
    den = torch.zeros((n, 1), dtype=x.dtype).index_add(0, y_idx, w)
    return num / den


# --------------------------------------------------------------------------------------
# A12 : packed <-> padded layouts
# --------------------------------------------------------------------------------------


def to_batch_padded(t, batch):
    """(N, ...) + sorted cloud ids -> zero padded (B, Nmax, ...) + bool mask (ref point_ops.py:358-381)."""
    bounds = segment_starts(batch, with_ends=True)
    lens = bounds[1:] - bounds[:-1]
    n_clouds, n_max = lens.numel(), int(lens.max().item())
    if n_clouds == 1:
        return t.unsqueeze(0), torch.ones((1, t.size(0)), dtype=torch.bool)
    local = torch.arange(batch.numel()) - bounds[batch]
    out = torch.zeros((n_clouds, n_max) + tuple(t.shape[1:]), dtype=t.dtype)
    mask = torch.zeros((n_clouds, n_max), dtype=torch.bool)
    out = out.index_put((batch, local), t)
    mask[batch, local] = True
    return out, mask


def padded_layout(t, batch):
    """ref point_ops.py:264-284 ``dense2padded_pyg``: (padded, mask, lengths, offsets)."""
    padded, mask = to_batch_padded(t, batch)
    lens = mask.sum(dim=-1)
    return padded, mask, lens, torch.cumsum(lens, 0)[:-1]


# --------------------------------------------------------------------------------------
# A11 : fixed-radius kNN (FRNN semantics), exhaustive search
# --------------------------------------------------------------------------------------

_BRUTE = None


def _brute_lib():
    global _BRUTE
    if _BRUTE is None:
        path = os.path.join(_HERE, "libccn_oracle.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle/libccn_oracle.so missing: run `make -C oracle` (or __graft_entry__.build())")
        lib = ctypes.CDLL(path)
        lib.ccn_oracle_frnn.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int64] * 4 + [ctypes.c_void_p] * 3
        lib.ccn_oracle_frnn.restype = None
        lib.ccn_oracle_knn.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int64] * 4 + [ctypes.c_void_p] * 2
        lib.ccn_oracle_knn.restype = None
        lib.ccn_oracle_ball_query.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int64] * 4 + [ctypes.c_float, ctypes.c_void_p]
        lib.ccn_oracle_ball_query.restype = None
        _BRUTE = lib
    return _BRUTE


def frnn_bruteforce(points1, points2, lengths1, lengths2, K, r, return_dists=False):
    """Semantics of ``frnn.frnn_grid_points`` as used at ref point_ops.py:459 (third-party, not
    vendored): for each query the <=K nearest points2 with d2 < r*r, ascending
    by (d2, index); int64 indices padded with -1; rows >= lengths1 are -1."""
    p1 = np.ascontiguousarray(points1.detach().numpy(), dtype=np.float32)
    p2 = np.ascontiguousarray(points2.detach().numpy(), dtype=np.float32)
    B, P1, _ = p1.shape
    P2 = p2.shape[1]
    l1 = np.ascontiguousarray(lengths1.numpy(), dtype=np.int64)
    l2 = np.ascontiguousarray(lengths2.numpy(), dtype=np.int64)
    if isinstance(r, (int, float)):
        rr = np.full(B, r, dtype=np.float32)
    else:
        rr = np.ascontiguousarray(torch.as_tensor(r, dtype=torch.float32).expand(B).numpy(), dtype=np.float32)
    idx = np.full((B, P1, K), -1, dtype=np.int64)
    d2 = np.full((B, P1, K), -1.0, dtype=np.float32)
    _brute_lib().ccn_oracle_frnn(p1.ctypes.data, p2.ctypes.data, l1.ctypes.data, l2.ctypes.data,
                                 B, P1, P2, K, rr.ctypes.data, idx.ctypes.data, d2.ctypes.data)
    if return_dists:
        return torch.from_numpy(idx), torch.from_numpy(d2)
    return torch.from_numpy(idx)


def knn_bruteforce(points1, points2, lengths1, lengths2, K):
    """Exact kNN with pytorch3d ``knn_points`` semantics (ref point_ops.py:91): ascending by
    (d2, index), -1 where a cloud has fewer than K points."""
    p1 = np.ascontiguousarray(points1.detach().numpy(), dtype=np.float32)
    p2 = np.ascontiguousarray(points2.detach().numpy(), dtype=np.float32)
    B, P1, _ = p1.shape
    P2 = p2.shape[1]
    l1 = np.ascontiguousarray(lengths1.numpy(), dtype=np.int64)
    l2 = np.ascontiguousarray(lengths2.numpy(), dtype=np.int64)
    idx = np.full((B, P1, K), -1, dtype=np.int64)
    d2 = np.full((B, P1, K), -1.0, dtype=np.float32)
    _brute_lib().ccn_oracle_knn(p1.ctypes.data, p2.ctypes.data, l1.ctypes.data, l2.ctypes.data,
                                B, P1, P2, K, idx.ctypes.data, d2.ctypes.data)
    return torch.from_numpy(idx)


def ball_query_bruteforce(points1, points2, lengths1, lengths2, K, r):
    """pytorch3d ``ball_query`` semantics (ref point_ops.py:81): first K in index order with d2 < r*r."""
    p1 = np.ascontiguousarray(points1.detach().numpy(), dtype=np.float32)
    p2 = np.ascontiguousarray(points2.detach().numpy(), dtype=np.float32)
    B, P1, _ = p1.shape
    l1 = np.ascontiguousarray(lengths1.numpy(), dtype=np.int64)
    l2 = np.ascontiguousarray(lengths2.numpy(), dtype=np.int64)
    idx = np.full((B, P1, K), -1, dtype=np.int64)
    _brute_lib().ccn_oracle_ball_query(p1.ctypes.data, p2.ctypes.data, l1.ctypes.data, l2.ctypes.data, B, P1,
                                       p2.shape[1], K, float(r), idx.ctypes.data)
    return torch.from_numpy(idx)


def ball_query_nd(points1, points2, lengths1, lengths2, K, r):
    """pytorch3d ``ball_query`` between D-dimensional rows (the feature-space search of ref dgcnn.py:114-127):
    d2 accumulated over the components in order, first K in index order with d2 < r*r."""
    B, P1, D = points1.shape
    idx = torch.full((B, P1, K), -1, dtype=torch.long)
    for b in range(B):
        q, s = points1[b, : int(lengths1[b])].detach(), points2[b, : int(lengths2[b])].detach()
        d2 = torch.zeros((q.size(0), s.size(0)), dtype=q.dtype)
        for d in range(D):
            diff = q[:, d, None] - s[None, :, d]
            d2 = d2 + diff * diff
        inside = d2 < r * r
        order = inside.cumsum(1)
        rows, cols = (inside & (order <= K)).nonzero(as_tuple=True)
        idx[b, rows, order[rows, cols] - 1] = cols
    return idx


def group_fixed_radius(p1, p2, batch1, batch2, knn, radius, return_dense=False, operation="knn", accel_knn=True):
    """ref point_ops.py:73-111 ``knn_ball_group_pytorch3d``: FRNN (accel_knn), exact kNN, or ball query (K=128)."""
    if radius is None and operation == "knn" and accel_knn:
        radius = 0.25                                  # quirk Q7
    q_pad, mask1, len1, off1 = padded_layout(p1, batch1)
    s_pad, mask2, len2, off2 = padded_layout(p2, batch2)
    if operation == "ball-group":
        query = ball_query_bruteforce if q_pad.size(2) == 3 else ball_query_nd
        nbr = query(q_pad, s_pad, len1, len2, 128, radius)
    elif accel_knn:
        if q_pad.size(2) != 3:
            raise ValueError("FRNN searches 3-D points only")
        nbr = frnn_bruteforce(q_pad, s_pad, len1, len2, knn, radius)
    else:
        nbr = knn_bruteforce(q_pad, s_pad, len1, len2, knn)
    if return_dense:
        return nbr, len2, mask1
    keep = (nbr != -1) & mask1[:, :, None]
    col = nbr.clone()
    col[1:] += off2.view(-1, 1, 1)
    qid = torch.arange(nbr.size(1)).view(1, -1, 1).expand_as(nbr).clone()
    qid[1:] += off1.view(-1, 1, 1)
    return qid[keep], col[keep]


# --------------------------------------------------------------------------------------
# A16 : the MLP block (PyG 2.3.0 ``MLP`` semantics)
# --------------------------------------------------------------------------------------


class _Norm(nn.Module):
    """PyG ``BatchNorm`` wrapper: the real BatchNorm1d lives in ``.module`` (state-dict key)."""

    def __init__(self, channels):
        super().__init__()
        self.module = nn.BatchNorm1d(channels)

    def forward(self, x):
        return self.module(x)


class MLP(nn.Module):
    """lin -> batch-norm -> act -> dropout per hidden layer; the last layer is a bare Linear when
    ``plain_last`` (ref base.py:32,64,90-125; mlp.py:13)."""

    def __init__(self, channel_list, dropout=0.0, act="relu", norm="batch_norm", plain_last=True, bias=True, **kwargs):
        super().__init__()
        assert norm == "batch_norm"
        self.channel_list, self.plain_last = list(channel_list), plain_last
        n_layers = len(channel_list) - 1
        if isinstance(dropout, (int, float)):           # PyG 2.3.0: per-layer list, the plain last layer is never dropped
            self.dropout = [float(dropout)] * n_layers
            if plain_last and n_layers:
                self.dropout[-1] = 0.0
        else:
            self.dropout = [float(d) for d in dropout]
            assert len(self.dropout) == n_layers
        assert act in ("relu", "leaky_relu")
        self.act = act
        self.lins = nn.ModuleList(nn.Linear(a, b, bias=bias) for a, b in zip(channel_list[:-1], channel_list[1:]))
        normed = channel_list[1:-1] if plain_last else channel_list[1:]
        self.norms = nn.ModuleList(_Norm(c) for c in normed)

    def forward(self, x):
        for i, (lin, norm) in enumerate(zip(self.lins, self.norms)):
            x = F.dropout(activation(norm(linear(x, lin)), self.act), p=self.dropout[i], training=self.training)
            if (MLP_DTYPE in ("bf16", "fp16") and STORE16 and ACT_TRACE is None and self.dropout[i] == 0.0
                    and (i + 1 < len(self.norms) or self.plain_last)):
                x = _RoundGradBF16.apply(x)          # (stored as bf16 rows in the product: see STORE16)
        if self.plain_last:
            x = F.dropout(linear(x, self.lins[-1]), p=self.dropout[-1], training=self.training)
        return x


# --------------------------------------------------------------------------------------
# A13 : PointNetConv2 on a bipartite edge list whose rows are grouped by destination
# --------------------------------------------------------------------------------------


def _segment_dense(values, dst, n_dst):
    """Edges sorted by destination -> (n_dst, Dmax, C) zero padded view + validity mask."""
    counts = torch.bincount(dst, minlength=n_dst)
    first = torch.cumsum(counts, 0) - counts
    slot = torch.arange(dst.numel()) - first[dst]
    dmax = int(counts.max().item()) if dst.numel() else 0
    dense = torch.zeros((n_dst, dmax) + tuple(values.shape[1:]), dtype=values.dtype)
    dense = dense.index_put((dst, slot), values)
    valid = torch.zeros((n_dst, dmax), dtype=torch.bool)
    valid[dst, slot] = True
    return dense, valid


# --------------------------------------------------------------------------------------
# max aggregation with a routing trace (test hook)
# --------------------------------------------------------------------------------------
# A max over neighbours is not differentiable where two slots tie to the last bit: a 1e-7 CPU/GPU rounding difference
# can move ONE argmax and re-route that entry's gradient.  The parity tests therefore (1) record, per aggregation and
# output entry, WHICH SOURCE POINT won (``MAX_TRACE["record"]``), compare that table with the product's and count the
# flipped entries, and (2) re-run the oracle with the product's table FORCED (``MAX_TRACE["force"]``): values are then
# taken from, and gradients routed to, the slot the GPU chose, so every gradient can be held to a tight tolerance.
MAX_TRACE = None        # None | {"record": [], "force": None | [tables...]}


class _RouteMax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, vals, slot):                   # vals (R, S, C), slot (R, C)
        ctx.save_for_backward(slot)
        ctx.s = vals.size(1)
        return vals.gather(1, slot[:, None, :]).squeeze(1)

    @staticmethod
    def backward(ctx, g):
        (slot,) = ctx.saved_tensors
        d = torch.zeros((g.size(0), ctx.s, g.size(1)), dtype=g.dtype)
        d.scatter_(1, slot[:, None, :], g[:, None, :])
        return d, None


def _traced_max(vals, valid, src_ids, fill, rows=None):
    """max over dim 1 of ``vals`` (R, S, C) restricted to ``valid`` (R, S); invalid slots count as ``fill`` (a finite
    constant that takes part in the max, ref dgcnn.py:187-189) or, with ``fill=None``, are ignored and rows without
    any valid slot give 0 (torch_scatter ``scatter_max``).  ``src_ids`` (R, S): the source point behind every slot.
    ``rows`` (R,) bool: the rows that exist in the packed layout (the trace tables hold only those)."""
    neg = torch.full((), float("-inf") if fill is None else fill, dtype=vals.dtype)
    masked = torch.where(valid[:, :, None], vals, neg)
    ids = torch.where(valid, src_ids, torch.full((), -1, dtype=src_ids.dtype))
    trace = MAX_TRACE
    if trace is None:
        out = masked.max(dim=1)[0]
    else:
        nat_slot = masked.max(dim=1)[1]                                  # (R, C), first maximum
        nat = ids.gather(1, nat_slot)
        if fill is None:
            nat = torch.where(valid.any(dim=1)[:, None], nat, torch.full((), -1, dtype=nat.dtype))
        trace["record"].append(nat if rows is None else nat[rows])
        if trace.get("force") is None:
            out = _RouteMax.apply(masked, nat_slot)
        else:
            want = trace["force"].pop(0)
            if rows is not None:
                full = torch.full_like(nat, -1)
                full[rows] = want
                want = full
            assert want.shape == nat.shape, (want.shape, nat.shape)
            hit = ids[:, :, None] == want[:, None, :]                    # (R, S, C)
            found = hit.any(dim=1)
            empty = ~valid.any(dim=1)
            ok = found | ((want == -1) & empty[:, None]) if fill is None else found
            assert bool(ok.all()), "forced routing names a source that is not among the oracle's neighbours"
            slot = hit.to(torch.uint8).argmax(dim=1)
            out = _RouteMax.apply(masked, slot)
            # how far below the true maximum the forced choices lie (0 where nothing flipped; ~1e-7 on a last-bit tie)
            gap = (masked.max(dim=1)[0] - out).detach()
            trace["max_gap"] = max(trace.get("max_gap", 0.0), float(gap.max()) if gap.numel() else 0.0)
    if fill is None:
        out = torch.where(valid.any(dim=1)[:, None], out, torch.zeros((), dtype=vals.dtype))
    return out


def _segment_max(msg, dst, src, n_dst):
    """scatter_max of ``msg`` over ``dst`` (0 for empty groups), ``src`` naming the source point of every edge."""
    dense, valid = _segment_dense(msg, dst, n_dst)
    ids, _ = _segment_dense(src.view(-1, 1), dst, n_dst)
    return _traced_max(dense, valid, ids.squeeze(-1), None)


def segment_softmax(src, dst, n_dst):
    """PyG ``softmax(src, index)``: exp(src - segment max) / (segment sum + 1e-16) per channel."""
    top = torch.full((n_dst, src.size(1)), float("-inf"), dtype=src.dtype)
    top = top.scatter_reduce(0, dst[:, None].expand_as(src), src.detach(), reduce="amax", include_self=True)
    e = (src - top[dst]).exp()
    tot = torch.zeros((n_dst, src.size(1)), dtype=src.dtype).index_add(0, dst, e) + 1e-16
    return e / tot[dst]


class PointNetConv2(nn.Module):
    """ref point_conv.py:12-93 with PyG 2.3.0 bipartite ``propagate`` semantics:
    x_j = x_src[edge[0]], pos_j = pos_src[edge[0]], pos_i = pos_dst[edge[1]], aggregate over edge[1]."""

    def __init__(self, local_nn, attend_nn=None, global_nn=None, aggr_type="max", normalize_radius=None):
        super().__init__()
        assert aggr_type in ("max", "attend", "mean", "weighted-sum")
        self.local_nn, self.attend_nn, self.global_nn = local_nn, attend_nn, global_nn
        self.aggr_type, self.normalize_radius = aggr_type, normalize_radius

    def forward(self, x_src, pos_src, pos_dst, src, dst):
        n_dst = pos_dst.size(0)
        rel = _feat(pos_src)[src] - _feat(pos_dst)[dst]
        if self.normalize_radius is not None:
            rel = rel / self.normalize_radius
        msg = rel if x_src is None else torch.cat([x_src[src], rel], dim=1)
        if self.local_nn is not None:
            msg = self.local_nn(msg)
        if self.aggr_type == "max":
            out = _segment_max(msg, dst, src, n_dst)
        elif self.aggr_type == "mean":
            cnt = torch.bincount(dst, minlength=n_dst).clamp(min=1).to(msg.dtype)
            out = torch.zeros((n_dst, msg.size(1)), dtype=msg.dtype).index_add(0, dst, msg) / cnt[:, None]
        elif self.aggr_type == "weighted-sum":
            w = torch.sigmoid(self.attend_nn(msg))
            out = torch.zeros((n_dst, msg.size(1)), dtype=msg.dtype).index_add(0, dst, msg * w)
        else:
            w = segment_softmax(self.attend_nn(msg), dst, n_dst)
            out = torch.zeros((n_dst, msg.size(1)), dtype=msg.dtype).index_add(0, dst, msg * w)
        if self.global_nn is not None:
            out = self.global_nn(out)
        return out


# --------------------------------------------------------------------------------------
# A7 users / A10 / A14 / A15 / A17 : step modules (same constructor + call surface as the reference)
# --------------------------------------------------------------------------------------


def _draw_u():
    return torch.rand(1)


class CurveFPS(nn.Module):
    def __init__(self, arclen_spacing=0.3):
        super().__init__()
        self.arclen_spacing = arclen_spacing

    def forward(self, pos, batch, point2curveidx, u=None):
        return curve_fps(pos, batch, point2curveidx, self.arclen_spacing, _draw_u() if u is None else u)


def farthest_point_indices(pos, batch, ratio, start=None):
    """ref point_ops.py:57-70 with pytorch3d ``sample_farthest_points`` semantics (App. C):
    per cloud ceil(len*ratio) samples, first = ``start[b]`` (the reference draws it at random),
    then repeatedly the point with the largest distance to the chosen set; sorted global indices."""
    bounds = segment_starts(batch, with_ends=True)
    out = []
    for b in range(bounds.numel() - 1):
        lo, hi = int(bounds[b]), int(bounds[b + 1])
        p = pos[lo:hi]
        n_keep = int(torch.ceil(torch.tensor([hi - lo]) * ratio))          # the reference's float32 product
        cur = int(start[b]) if start is not None else int(torch.randint(hi - lo, (1,)))
        d = torch.full((hi - lo,), float("inf"), dtype=pos.dtype)
        chosen = []
        for _ in range(n_keep):
            chosen.append(cur)
            diff = p - p[cur]
            # explicit (dx*dx + dy*dy) + dz*dz in float32: FPS is chaotic, the HIP kernel uses the same expression
            d2 = (diff[:, 0] * diff[:, 0] + diff[:, 1] * diff[:, 1]) + diff[:, 2] * diff[:, 2]
            d = torch.minimum(d, d2)
            cur = int(torch.argmax(d))
        out.append(torch.tensor(chosen, dtype=torch.long) + lo)
    return torch.sort(torch.cat(out))[0]


def voxel_fps(pos, batch, voxel_size, rnd):
    """ref fps_ops.py:51-60 VoxelFPS: per occupied (cloud, voxel) the point closest to the voxel corner after a
    random perturbation ``rnd * voxel_size / 4``; result in the lexicographic voxel order of torch.unique."""
    scaled = pos / voxel_size
    vox = torch.floor(scaled).long()
    cells = torch.cat([batch.view(-1, 1), vox], dim=-1)
    _, cell_of = torch.unique(cells, dim=0, return_inverse=True)
    score = torch.linalg.norm(vox - scaled, dim=-1) + rnd * voxel_size / 4
    order = np.lexsort((np.arange(pos.size(0)), score.numpy(), cell_of.numpy()))   # by cell, then score, then index
    cells_sorted = cell_of.numpy()[order]
    first = np.concatenate([[True], cells_sorted[1:] != cells_sorted[:-1]])
    return torch.from_numpy(order[first].astype(np.int64))


class SAModule(nn.Module):
    """ref pointnet2.py:33-78."""

    def __init__(self, ratio, r, nn, k, curve_fps_arclen=None, voxel_size=None, downsample_type="random",
                 attend_nn=None, aggr_type="max", normalize_radius=False, use_fast_knn=True, **kwargs):
        super().__init__()
        assert downsample_type in ("curve-fps", "random", "fps", "voxel")
        self.ratio, self.r, self.knn = ratio, r, k
        self.downsample_type, self.use_fast_knn = downsample_type, use_fast_knn
        self.curve_fps_arclen, self.voxel_size = curve_fps_arclen, voxel_size
        self.conv = PointNetConv2(nn, attend_nn=attend_nn, aggr_type=aggr_type,
                                  normalize_radius=r if normalize_radius else None)

    def forward(self, x, pos, batch, point2curveidx=None, **kwargs):
        if "sample_idx" in kwargs and kwargs["sample_idx"] is not None:
            idx = kwargs["sample_idx"]
        elif self.downsample_type == "random":
            idx = torch.sort(torch.randperm(pos.size(0))[: int(pos.size(0) * self.ratio)])[0]
        elif self.downsample_type == "curve-fps":
            idx = curve_fps(pos, batch, point2curveidx, self.curve_fps_arclen, _draw_u())
        elif self.downsample_type == "fps":
            idx = farthest_point_indices(pos, batch, self.ratio)
        else:
            idx = voxel_fps(pos, batch, self.voxel_size, torch.rand(pos.size(0)))
        row, col = group_fixed_radius(pos[idx], pos, batch[idx], batch, self.knn, self.r,
                                      operation="knn" if self.use_fast_knn else "ball-group")
        x = self.conv(x, pos, pos[idx], col, row)
        p2c = None if point2curveidx is None else point2curveidx[idx]
        return x, pos[idx], batch[idx], p2c


class CurveSAModule(nn.Module):
    """ref pointnet2.py:146-181."""

    def __init__(self, ratio, r, nn, curve_fps_arclen=None, use_curve_fps=False, global_nn=None, attend_nn=None,
                 with_xyz=False, aggr_type="max", normalize_radius=False, **kwargs):
        super().__init__()
        self.ratio, self.r, self.curve_fps_arclen = ratio, r, curve_fps_arclen
        self.use_curve_fps, self.with_xyz = use_curve_fps, with_xyz
        self.conv = PointNetConv2(nn, attend_nn=attend_nn, global_nn=global_nn, aggr_type=aggr_type,
                                  normalize_radius=r if normalize_radius else None)

    def forward(self, x, pos, batch, point2curveidx, **kwargs):
        if self.with_xyz:
            x = _feat(pos[:, :3]) if x is None else torch.cat([x, _feat(pos[:, :3])], dim=1)
        if "sample_idx" in kwargs and kwargs["sample_idx"] is not None:
            idx = kwargs["sample_idx"]
        elif self.use_curve_fps:
            idx = curve_fps(pos, batch, point2curveidx, self.curve_fps_arclen, _draw_u())
        else:
            idx = farthest_point_indices(pos, batch, self.ratio)
        row, col = curve_radius_group(pos, idx, point2curveidx, batch, self.r)
        x = self.conv(x, pos, pos[idx], col, row)
        return x, pos[idx], batch[idx], point2curveidx[idx], None, idx


def knn_interpolate(x, pos_x, pos_y, batch_x, batch_y, k):
    """ref point_ops.py:293-341 (exact kNN, inverse squared distance weights)."""
    with torch.no_grad():
        q_pad, mask1, len1, off1 = padded_layout(pos_y, batch_y)
        s_pad, mask2, len2, off2 = padded_layout(pos_x, batch_x)
        nbr = knn_bruteforce(q_pad, s_pad, len1, len2, k)
        keep = (nbr != -1) & mask1[:, :, None]
        col = nbr.clone()
        col[1:] += off2.view(-1, 1, 1)
        qid = torch.arange(nbr.size(1)).view(1, -1, 1).expand_as(nbr).clone()
        qid[1:] += off1.view(-1, 1, 1)
        y_idx, x_idx = qid[keep], col[keep]
        diff = pos_x[x_idx] - pos_y[y_idx]
        w = (1.0 / torch.clamp((diff * diff).sum(dim=-1, keepdim=True), min=1e-16)).to(x.dtype)
    n = pos_y.size(0)
    num = torch.zeros((n, x.size(1)), dtype=x.dtype).index_add(0, y_idx, x[x_idx] * w)
    den = torch.zeros((n, 1), dtype=x.dtype).index_add(0, y_idx, w)
    return num / den


def _fp_concat(x, x_skip, pos_skip, with_xyz):
    parts = [x]
    if x_skip is not None:
        parts.append(x_skip)
    if with_xyz:
        parts.append(_feat(pos_skip[:, :3]))
    return torch.cat(parts, dim=1)


class FPModule(nn.Module):
    """ref pointnet2.py:119-143."""

    def __init__(self, k, nn, with_xyz=False):
        super().__init__()
        self.k, self.nn, self.with_xyz = k, nn, with_xyz

    def forward(self, x, pos, batch, x_skip, pos_skip, batch_skip, point2curveidx=None, point2curveidx_skip=None, **kwargs):
        x = knn_interpolate(x, pos, pos_skip, batch, batch_skip, self.k)
        return self.nn(_fp_concat(x, x_skip, pos_skip, self.with_xyz)), pos_skip, batch_skip, point2curveidx_skip


class CurveFPModule(FPModule):
    """ref pointnet2.py:184-205."""

    def forward(self, x, idx, x_skip, pos_skip, batch_skip, point2curveidx_skip=None, **kwargs):
        x = curve_interpolate(x, idx, pos_skip, batch_skip, point2curveidx_skip, self.k)
        return self.nn(_fp_concat(x, x_skip, pos_skip, self.with_xyz)), pos_skip, batch_skip, point2curveidx_skip


class SGCNNLayer(nn.Module):
    """ref dgcnn.py:130-266, dense FRNN path ``forward_fast`` (quirk Q4: the MLP and its batch
    statistics run over all B*Nmax*(K+1) rows, masking afterwards)."""

    def __init__(self, nn, k, aggr="max", r=1.0, num_workers=1, with_xyz=False, attend_nn=None, aggr_type="max",
                 use_sparse_feat_agg=False, use_fast_knn=True, **kwargs):
        super().__init__()
        assert aggr_type in ("max", "attend", "mean", "weighted-sum")
        self.nn, self.k, self.r, self.with_xyz = nn, k, r, with_xyz
        self.attend_nn, self.aggr_type, self.use_sparse_feat_agg = attend_nn, aggr_type, use_sparse_feat_agg
        self.use_fast_knn = use_fast_knn

    def forward(self, x, pos, batch, point2curveidx=None, **kwargs):
        if self.with_xyz:
            x = _feat(pos) if x is None else torch.cat([x, _feat(pos)], dim=1)
        if self.use_sparse_feat_agg:
            # ref dgcnn.py:209-246 forward_slow: message nn([x_i, x_j - x_i]) over the edge list, aggregate per query
            row, col = group_fixed_radius(pos, pos, batch, batch, self.k, self.r, accel_knn=self.use_fast_knn)
            msg = self.nn(torch.cat([x[row], x[col] - x[row]], dim=-1))
            n = x.size(0)
            if self.aggr_type == "max":
                out = _segment_max(msg, row, col, n)
            else:
                w = segment_softmax(self.attend_nn(msg), row, n)
                out = torch.zeros((n, msg.size(1)), dtype=msg.dtype).index_add(0, row, msg * w)
            return out, pos, batch, point2curveidx
        nbr, len2, mask1 = group_fixed_radius(pos, pos, batch, batch, self.k, self.r, return_dense=True)
        B, N = nbr.shape[:2]
        me = torch.arange(N).view(1, N, 1).expand(B, N, 1)
        nbr = torch.cat([me, nbr], dim=2)
        xp = to_batch_padded(x, batch)[0]
        gathered = torch.gather(xp[:, :, None, :].expand(-1, -1, nbr.size(2), -1), 1,
                                nbr.clamp(min=0)[..., None].expand(-1, -1, -1, xp.size(-1)))
        gathered = torch.where((nbr >= 0)[..., None], gathered, torch.zeros((), dtype=x.dtype))
        edge = torch.cat([gathered, gathered[:, :, 0:1, :] - gathered], dim=-1)
        f = self.nn(edge.reshape(-1, edge.size(-1))).view(B, N, self.k + 1, -1)
        mask = (nbr != -1) & mask1[:, :, None]
        if self.aggr_type == "max":
            first = torch.cumsum(len2, 0) - len2                                   # packed index of every cloud's first point
            f = _traced_max(f.reshape(B * N, self.k + 1, -1), mask.reshape(B * N, -1),
                            (nbr + first.view(B, 1, 1)).reshape(B * N, -1), -1e2, rows=mask1.reshape(-1)).view(B, N, -1)
        elif self.aggr_type == "mean":
            f = torch.where(mask[..., None], f, torch.zeros((), dtype=f.dtype)).sum(dim=2) / mask.sum(dim=2)[..., None]
        elif self.aggr_type == "weighted-sum":
            a = torch.sigmoid(self.attend_nn(f.reshape(B * N * (self.k + 1), -1)).view(B, N, self.k + 1, -1))
            a = torch.where(mask[..., None], a, torch.zeros((), dtype=f.dtype))
            a = a / torch.clamp(a.sum(dim=2, keepdim=True), min=1e-3)
            f = (f * a).sum(dim=2)
        else:
            a = self.attend_nn(f.reshape(B * N * (self.k + 1), -1)).view(B, N, self.k + 1, -1)
            a = torch.where(mask[..., None], a, torch.full((), -5e2, dtype=f.dtype))
            f = (f * F.softmax(a, dim=2)).sum(dim=2)
        return f[mask1], pos, batch, point2curveidx


class DGCNNLayer(nn.Module):
    """ref dgcnn.py:16-111 (step "dgcnn"): neighbours searched between feature vectors with the FRNN default of
    ``knn_ball_group_pytorch3d`` (radius 0.25), message nn([x_i, x_j - x_i]), max per query (0 for empty groups)."""

    def __init__(self, nn, k, aggr="max", num_workers=1, with_xyz=False, **kwargs):
        super().__init__()
        self.nn, self.k, self.r, self.with_xyz, self.operation = nn, k, None, with_xyz, "knn"

    def forward(self, x, pos, batch, point2curveidx=None, **kwargs):
        if self.with_xyz:
            x = _feat(pos) if x is None else torch.cat([x, _feat(pos)], dim=1)
        row, col = group_fixed_radius(x.detach(), x.detach(), batch, batch, self.k, self.r, operation=self.operation)
        msg = self.nn(torch.cat([x[row], x[col] - x[row]], dim=-1))
        out = _segment_max(msg, row, col, x.size(0))
        return out, pos, batch, point2curveidx


class DGCNNLayerRadius(DGCNNLayer):
    """ref dgcnn.py:114-127 (step "dgcnn-rad"): ball query (K=128, index order) between feature vectors."""

    def __init__(self, nn, r, aggr="max", num_workers=1, with_xyz=False, **kwargs):
        super().__init__(nn, None, aggr, num_workers, with_xyz)
        self.r, self.operation = r, "ball-group"


class GlobalSAModule(nn.Module):
    """ref pointnet2.py:81-116: per-cloud max (or mean) pooling of nn([x, pos])."""

    def __init__(self, nn, **kwargs):
        super().__init__()
        self.nn, self.pooling = nn, kwargs.get("pooling", "max")

    def forward(self, x, pos, batch, point2curveidx=None, **kwargs):
        f = self.nn(torch.cat([x, _feat(pos)], dim=1))
        n_clouds = int(batch.max().item()) + 1
        if self.pooling == "max":
            f = _segment_max(f, batch, torch.arange(f.size(0)), n_clouds)
        else:
            cnt = torch.bincount(batch, minlength=n_clouds).to(f.dtype)
            f = torch.zeros((n_clouds, f.size(1)), dtype=f.dtype).index_add(0, batch, f) / cnt[:, None]
        first = torch.cat([torch.zeros(1, dtype=torch.long), segment_starts(batch)])
        return f, pos[first], batch[first], None if point2curveidx is None else point2curveidx[first]


class SharedMLP(nn.Module):
    """ref mlp.py:5-22."""

    def __init__(self, dims, use_bias=False, with_xyz=False, act="leaky_relu", **kwargs):
        super().__init__()
        self.mlp = MLP(dims, dropout=kwargs.get("dropout", 0.0), norm=kwargs.get("norm", "batch_norm"),
                       plain_last=kwargs.get("plain_last", True), act=act, bias=use_bias)
        self.with_xyz = with_xyz

    def forward(self, x, pos, batch, point2curveidx=None, **kwargs):
        if self.with_xyz:
            x = _feat(pos) if x is None else torch.cat([x, _feat(pos)], dim=1)
        return self.mlp(x), pos, batch, point2curveidx


class SkipConnect(nn.Module):
    """ref skip_connect.py:6-14."""

    def __init__(self, nn, num_skips=1):
        super().__init__()
        self.num_skips, self.nn = num_skips, nn

    def forward(self, xs, pos, batch, point2curveidx=None, **kwargs):
        return self.nn(torch.cat(xs, dim=1)), pos, batch, point2curveidx


# --------------------------------------------------------------------------------------
# A18 : model assembly from the reference's config dict
# --------------------------------------------------------------------------------------

_DOWN = ("sa", "sa-geo", "sa-global", "pt-transition-down")


class ModelBase(nn.Module):
    """ref base.py:16-215 (step table, dimension inference, skip-state machine, final MLP)."""

    def __init__(self, in_dim, n_out, steps, feat_dims, out_mlp=dict(), **kwargs):
        super().__init__()
        self.in_dim, self.n_out = in_dim, n_out
        self.use_bias = kwargs.get("use_bias", False)
        self.version = kwargs.get("version", 2.0)
        self.step_names = list(steps)
        self.skip_connect_state_store = list(kwargs.get("skip_connect_state_store", []))
        self.steps = nn.ModuleList()
        for i, name in enumerate(steps):
            kw = dict(kwargs)
            if isinstance(name, dict):
                kw.update(name)
                name = kw.pop("step_name")
                self.step_names[i] = name
            kw["with_xyz"] = kw.get("with_xyz", False)
            self.steps.append(self._make(i, name, self._dims(i, name, feat_dims, in_dim, kw["with_xyz"]), **kw))
        tail = {"dropout": 0.5, "norm": "batch_norm", "plain_last": True}
        spec = copy.deepcopy(out_mlp)
        if isinstance(spec, dict):
            hidden = spec.pop("dims") or []
            tail.update(spec)
        else:
            hidden = spec
        dims = [feat_dims[-1][-1]] + list(hidden) + [n_out]
        if tail.pop("with_seg_category", False):
            dims[0] += 64
            self.lin_categorical = MLP([16, 64, 64])
        self.mlp = nn.Identity() if tail.pop("identity", False) else MLP(dims, bias=self.use_bias, **tail)

    @staticmethod
    def _dims(i, name, feat_dims, in_dim, xyz):
        prev = in_dim if i == 0 else feat_dims[i - 1][-1]
        if name in ("dgcnn", "sgcnn"):
            head = [in_dim * 2] if i == 0 else [2 * (prev + 3 * xyz)]
        elif name in ("sa", "sa-global", "sa-geo"):
            head = [in_dim + 3 * xyz] if i == 0 else [prev + 3 + 3 * xyz]
        elif name in ("skip-connect", "fp", "fp-geo") and i != 0:
            head = []
        elif name in ("mlp", "conv1d-fast-v1", "conv1d-fast-v2") or i == 0:
            head = [in_dim] if i == 0 else [prev + 3 * xyz]
        else:
            raise NotImplementedError("No Module Named >> %s" % name)
        return head + list(feat_dims[i])

    def _attend(self, dims, kw, halve):
        if kw.get("aggr_type") not in ("attend", "weighted-sum"):
            return None
        c = dims[-1]
        mid = c // 2 if (halve and self.version == 2.0) else c
        return MLP([c, mid, c], act="leaky_relu", bias=self.use_bias)

    def _make(self, i, name, dims, **kw):
        b = self.use_bias
        if name == "sa":
            return SAModule(kw["ratios"][i], kw["radii"][i], MLP(dims, bias=b), attend_nn=self._attend(dims, kw, True),
                            k=kw["knn"][i], **kw)
        if name == "sgcnn":
            return SGCNNLayer(MLP(dims, bias=b), kw["knn"][i], r=kw["radii"][i], attend_nn=self._attend(dims, kw, False), **kw)
        if name == "sa-global":
            return GlobalSAModule(MLP(dims, bias=b), **kw)
        if name == "dgcnn":
            return DGCNNLayer(MLP(dims, bias=b), kw["knn"][i], with_xyz=kw["with_xyz"])
        if name == "dgcnn-rad":
            return DGCNNLayerRadius(MLP(dims, bias=b), kw["radii"][i], with_xyz=kw["with_xyz"])
        if name == "sa-geo":
            return CurveSAModule(kw["ratios"][i], kw["radii"][i], MLP(dims, act="leaky_relu", bias=b),
                                 attend_nn=self._attend(dims, kw, False), **kw)
        if name == "conv1d-fast-v1":
            return SymmetricCurve1DConvFastV1(dims, kw["kernel_sizes"][i], with_xyz=kw["with_xyz"], with_diff=kw.get("with_diff", False))
        if name == "conv1d-fast-v2":
            return SymmetricCurve1DConvV2(dims, kw["kernel_sizes"][i], with_xyz=kw["with_xyz"], with_diff=kw.get("with_diff", False))
        if name == "skip-connect":
            return SkipConnect(MLP(dims, act="leaky_relu", bias=b), kw["num_skips"][i])
        if name == "fp":
            return FPModule(kw["knn"][i], MLP(dims, bias=b), with_xyz=kw["with_xyz"])
        if name == "fp-geo":
            return CurveFPModule(kw["knn"][i], MLP(dims, act="leaky_relu", bias=b), with_xyz=kw["with_xyz"])
        if name == "mlp":
            return SharedMLP(dims, **kw)
        raise NotImplementedError("Have not implemented step %s yet!" % name)

    def forward(self, data, **kwargs):
        x, pos, batch, p2c = data.x, data.pos, data.batch, data.curve_idxs
        x = None if x is None else _feat(x)
        if hasattr(data, "labels"):
            kwargs["shapenet-categories"] = data.labels
        hist = {"x": [x], "pos": [pos], "batch": [batch], "p2c": [p2c], "idx": []}
        keep_prop, keep_down = [], []
        cloud_of_point = batch
        for i, (name, step) in enumerate(zip(self.step_names, self.steps)):
            if name in ("fp", "fp-geo"):
                j = keep_down.pop()
                skip_x = hist["x"][j] if hist["x"][j] is not None else _feat(hist["pos"][j])
                if name == "fp":
                    out = step(x, pos, batch, skip_x, hist["pos"][j], hist["batch"][j], p2c, hist["p2c"][j], **kwargs)
                else:
                    out = step(x, hist["idx"][j], skip_x, hist["pos"][j], hist["batch"][j], hist["p2c"][j], **kwargs)
            elif name == "skip-connect":
                take = keep_prop[-step.num_skips:]
                del keep_prop[-step.num_skips:]
                xs = [x] + [hist["x"][j] if hist["x"][j] is not None else _feat(hist["pos"][j]) for j in take]
                out = step(xs, pos, batch, p2c, **kwargs)
            else:
                out = step(x, pos, batch, p2c, **kwargs)
            x, pos, batch, p2c = out[:4]
            hist["x"].append(x); hist["pos"].append(pos); hist["batch"].append(batch); hist["p2c"].append(p2c)
            hist["idx"].append(out[5] if len(out) > 5 else None)
            if name in self.skip_connect_state_store:
                keep_prop.append(i)
            if name in _DOWN:
                keep_down.append(i)
        if "shapenet-categories" in kwargs and hasattr(self, "lin_categorical"):
            cats = self.lin_categorical(_feat(F.one_hot(kwargs["shapenet-categories"], num_classes=16).float()))
            x = torch.cat([x, cats[cloud_of_point]], dim=1)
        return self.mlp(x)


def segmentation_loss(logits, target):
    """Mean negative log-likelihood over points (ref src/run/kitti_seg.py:184-192)."""
    return F.nll_loss(F.log_softmax(logits, dim=-1), target)


# --------------------------------------------------------------------------------------
# harness rows: Lovasz-softmax loss and the dataset-side curve splitters
# --------------------------------------------------------------------------------------

def lovasz_gradient(fg_sorted):
    """Gradient of the Lovasz extension w.r.t. the sorted errors (ref src/models/utils/lovasz_losses.py:19-31)."""
    total = fg_sorted.sum()
    inter = total - fg_sorted.float().cumsum(0)
    union = total + (1 - fg_sorted).float().cumsum(0)
    jac = 1.0 - inter / union
    if fg_sorted.numel() > 1:
        jac[1:] = jac[1:] - jac[:-1]
    return jac


def lovasz_softmax_flat(probas, labels):
    """ref lovasz_losses.py:174-202 with classes='present': mean over the classes that occur in ``labels`` of
    <errors sorted descending, lovasz_gradient(foreground in that order)>."""
    if probas.numel() == 0:
        return probas * 0.0
    per_class = []
    for c in range(probas.size(1)):
        fg = (labels == c).float()
        if fg.sum() == 0:
            continue
        err = (fg - probas[:, c]).abs()
        err_sorted, order = torch.sort(err, 0, descending=True)
        per_class.append(torch.dot(err_sorted, lovasz_gradient(fg[order])))
    return sum(per_class) / len(per_class)


def seg_loss_kitti(pred, gt, ignore=0, use_lovasz=False, class_weights=None):
    """ref src/run/kitti_seg.py:184-202: (loss, per-point NLL); mean over ALL points of the ignore-masked NLL,
    plus 2x the Lovasz-softmax loss over the non-ignored points."""
    logp = F.log_softmax(pred, dim=-1)
    if class_weights is None:
        per_point = F.nll_loss(logp, gt, reduction="none", ignore_index=ignore)
    else:
        assert ignore == 0
        w = torch.cat([torch.zeros(1, dtype=class_weights.dtype), class_weights], dim=0)
        per_point = F.nll_loss(logp, gt, reduction="none", weight=w)
    loss = per_point.mean()
    if use_lovasz:
        keep = gt != ignore
        loss = loss + 2 * lovasz_softmax_flat(F.softmax(pred, dim=-1)[keep], gt[keep]).mean()
    return loss, per_point


def split_curves(points, beam_idxs=None, thresh=0.08):
    """Curve ids of a sweep in acquisition order (ref src/data/kitti_dataset.py:73-92 with beam_idxs=None,
    src/data/nuscenes_dataset.py:101-118 after the beam sort): a new curve starts where the beam changes or where
    the fp64 edge length exceeds thresh * sqrt(xy-radius of the later point) (fp32 right-hand side).  int64 (N,)."""
    edges = points[1:].double() - points[:-1].double()
    edge_len = torch.linalg.norm(edges, dim=-1)
    radius = torch.linalg.norm(points[1:, :2], dim=-1)
    split = edge_len > (thresh * torch.sqrt(radius))
    if beam_idxs is not None:
        split = split | ((beam_idxs[1:] - beam_idxs[:-1]) != 0)
    return torch.cat([torch.zeros(1, dtype=torch.long), torch.cumsum(split, dim=0)], dim=0)


def get_curves_nuscenes(points, beam_idxs, labels, reflectance, thresh=0.08):
    """ref nuscenes_dataset.py:91-118: stable sort by beam, split, and the inverse permutation."""
    order = torch.sort(beam_idxs, stable=True)[1]
    inverse = torch.empty_like(order)
    inverse[order] = torch.arange(points.size(0)).to(order)
    points, beam_idxs, labels, reflectance = points[order], beam_idxs[order], labels[order], reflectance[order]
    return points, split_curves(points, beam_idxs, thresh), labels, reflectance, inverse
