"""Host-side operators: thin wrappers + ``torch.autograd.Function`` classes over the HIP C-ABI.

PyTorch supplies device memory, streams and autograd bookkeeping only; every arithmetic step on
the hot path is a ``ccn_*`` kernel launch.  Reference call sites are cited per operator.
"""
import ctypes
import os
import threading

import torch

from ._lib import call, lib, ptr, require_gpu, workspace

ACT = {None: 0, "none": 0, "relu": 1, "leaky_relu": 2}
LEAKY_SLOPE = 0.01  # torch.nn.LeakyReLU default, the only slope the reference uses

# Test hook: when set to a list, every max aggregation appends an (rows, C) int64 table naming, per output entry, the
# SOURCE POINT whose value won (-1: the -1e2 fill of an empty slot / an empty group).  The parity tests compare these
# tables with the oracle's to count arg-max flips on last-bit ties, and feed them back to the oracle so that gradients
# are compared along identical routes.
MAX_TRACE = None


# Test hook: when set to a list, every BatchNorm + activation layer appends the sign table (z > 0, bool, on the CPU) of its
# output in forward order; ACT_ROW_MAP (set by the compact-row SGCNN path) re-indexes the rows to the reference's dense
# B*Nmax*(K+1)-row layout first.  The parity tests make the oracle take the same ReLU / LeakyReLU slopes (see
# oracle.torch_ref.ACT_TRACE) so that gradients are compared along identical routes.
ACT_TRACE = None
ACT_ROW_MAP = None


def _trace_act(z, act):
    if ACT_TRACE is None or act == 0:
        return
    m = z > 0
    if ACT_ROW_MAP is not None:
        m = m[ACT_ROW_MAP]
    ACT_TRACE.append(m.cpu())


def _trace_max(arg, first, ids):
    """arg (rows, C) position of the winner inside its group (-1: none); first (rows,) start of each group in ``ids``."""
    pos = first.long()[:, None] + arg.long().clamp(min=0)
    MAX_TRACE.append(torch.where(arg >= 0, ids.long()[pos.clamp(max=max(ids.numel() - 1, 0))],
                                 torch.full((), -1, dtype=torch.int64, device=arg.device)).cpu())


def _mat(t):
    """float32 2-D row-major matrix, possibly with a padded leading dimension (a column-slice view)."""
    require_gpu(t)
    if t.dtype != torch.float32:
        raise TypeError("expected float32, got %s" % t.dtype)
    if t.dim() == 2 and t.size(1) > 0 and t.stride(1) == 1 and t.stride(0) >= t.size(1):
        return t
    return t.contiguous()


def _ld(t):
    """leading dimension (elements between consecutive rows)"""
    return max(t.stride(0), t.size(1), 1) if t.dim() == 2 else t.size(-1)


def _rows(rows, cols, device, zero=False):
    """(rows, cols) float32 matrix whose rows start 16-byte aligned (leading dimension padded to a multiple
    of 4), so that the GEMM loaders can use 16-byte loads for widths like 67, 131, 134 or 259."""
    ld = (cols + 3) // 4 * 4
    alloc = torch.zeros if zero else torch.empty
    buf = alloc((rows, ld), dtype=torch.float32, device=device)
    return buf if ld == cols else buf[:, :cols]


def _rows_halo(rows, cols, h, device, init=True):
    """(rows, cols) matrix for the implicit-GEMM curve convolution (ccn_conv_rows_*): ``h`` zeroed halo rows in front of and
    behind it in the same allocation, the leading dimension padded with ZERO columns to a multiple of 4 (to a multiple of 32
    from 64 channels on, so that taps * ld is a whole number of 32-deep GEMM slices).  The buffer is remembered on the view
    (``_ccn_halo``): a consumer that does not find it makes its own halo copy."""
    ld = (cols + 31) // 32 * 32 if cols >= 64 else (cols + 3) // 4 * 4
    buf = torch.empty((rows + 2 * h, ld), dtype=torch.float32, device=device)
    if init:
        if h:
            buf[:h].zero_()
            buf[rows + h:].zero_()
        if ld != cols:
            buf[:, cols:].zero_()
    view = buf[h:h + rows, :cols]
    view._ccn_halo = (buf, h)
    return view


def _halo_of(x, h):
    """(buffer, halo rows) when ``x`` is a _rows_halo view with at least h halo rows, else None."""
    info = getattr(x, "_ccn_halo", None)
    if info is None:
        return None
    buf, hh = info
    if (hh < h or x.dim() != 2 or x.stride(0) != buf.stride(0) or x.stride(1) != 1 or buf.size(0) != x.size(0) + 2 * hh
            or x.data_ptr() != buf.data_ptr() + hh * buf.stride(0) * 4):
        return None
    return buf, hh


def cat_cols(parts):
    """torch.cat(parts, dim=1) into an aligned-row buffer (autograd: plain column slices)."""
    parts = [p for p in parts if p is not None]
    if len(parts) == 1:
        return parts[0]
    return _CatCols.apply(*parts)


class _CatCols(torch.autograd.Function):
    @staticmethod
    def forward(ctx, *parts):
        widths = [p.size(1) for p in parts]
        out = _rows(parts[0].size(0), sum(widths), parts[0].device)
        off = 0
        for p, w in zip(parts, widths):
            out[:, off:off + w].copy_(p)
            off += w
        ctx.widths = widths
        return out

    @staticmethod
    def backward(ctx, g):
        outs, off = [], 0
        for w in ctx.widths:
            outs.append(g[:, off:off + w])
            off += w
        return tuple(outs)


def _pos(t):
    """(N, 3) positions, densely packed (the index kernels read xyz with stride 3)."""
    return _mat(t).contiguous()


def _i64(t):
    require_gpu(t)
    return t.to(torch.int64).contiguous()


# --------------------------------------------------------------------------------------
# Data-dependent element counts (sampled points, edges, compact rows, curves ...)
# --------------------------------------------------------------------------------------
# The reference turns every such count into a host integer with a device -> host synchronisation (``torch.where`` in
# ``batch2ptr`` point_ops.py:50, boolean-mask flattening :101-107, ``fps_ops.py:31-33``); by default this package does the same
# with one small read-back per count (``_count``).  ``COUNTS`` switches that:
#   * ``CountRecorder``: read back as usual and LOG every count in program order (the calibration pass of
#     ``graph.CapturedWholeForward``);
#   * ``CountBounds``: no read-back at all -- the host integer is a CAPACITY fixed beforehand (the calibrated count plus
#     head-room), every buffer is allocated for it, every kernel launched for it; the true count stays on the device, where
#     the producing kernel left it for its consumers (CSR offsets, group pointers, per-cloud lengths), the unused tail of an
#     index list is pre-filled with a harmless entry, and ``overflow`` -- a device flag -- is raised when a true count exceeds
#     its capacity (or a sortedness check fails).  One read-back of that flag per forward replaces all the others, and the
#     whole forward, geometry included, is hipGraph-capturable ("device-side counts").
COUNTS = None
_COUNTS_THREAD = None       # the thread a counts_scope() was entered on: other threads (ModelBase.prepare_async's worker) see None


def counts():
    """The count resolver in force FOR THE CALLING THREAD.  A resolver installed by ``counts_scope`` belongs to the thread that
    entered the scope -- a geometry worker (``ModelBase.prepare_async``) running meanwhile keeps reading its counts back and never
    consumes the scope's capacities out of step.  A resolver assigned to ``ops.COUNTS`` directly is process-wide."""
    if COUNTS is None or _COUNTS_THREAD is None or _COUNTS_THREAD == threading.get_ident():
        return COUNTS
    return None


class counts_scope:
    """``with counts_scope(resolver):`` -- install a CountRecorder / CountBounds for the enclosed passes of THIS thread."""

    def __init__(self, resolver):
        self.resolver = resolver

    def __enter__(self):
        global COUNTS, _COUNTS_THREAD
        if COUNTS is not None:
            raise RuntimeError("a count resolver is already installed (counts_scope does not nest)")
        COUNTS, _COUNTS_THREAD = self.resolver, threading.get_ident()
        return self.resolver

    def __exit__(self, et, ev, tb):
        global COUNTS, _COUNTS_THREAD
        COUNTS = _COUNTS_THREAD = None
        return False


class CountRecorder:
    """Ordinary counts (read back), logged; the samplers' host-side random draws logged as well -- or, given ``replay``
    (the ``draws`` of an earlier recorder), replayed instead of drawn."""

    def __init__(self, replay=None):
        self.log = []               # (what, value) in program order
        self.draws = []             # host tensors, in program order
        self.replay = None if replay is None else list(replay)

    def resolve(self, values, whats):
        for w, v in zip(whats, values):
            self.log.append((w, int(v)))
        return [int(v) for v in values]

    def draw(self, make, fit, device):
        t = self.replay.pop(0) if self.replay is not None else make()
        self.draws.append(t.clone())
        return t if device is None else t.to(device)


class CountBounds:
    """``caps``: the capacities, ``[(what, capacity)]`` in program order -- or None to CALIBRATE: the first pass then reads
    every true count back (as the synchronous path does), turns it into a capacity (``headroom`` x count, rounded up to 64)
    and USES that capacity at once, so that whatever the slack of one stage adds to the counts of the next -- the phantom
    points of a padded sample list have neighbours, rows and edges of their own -- is part of the later calibrations.
    After ``rewind()`` the same object replays the capacities without any read-back; ``rewind(verifying=True)`` replays them
    but reads every count back first and raises ``Exceeded`` BEFORE an overflowing count is used -- the safe way to find out
    whether another batch fits (a count past its capacity leaves later stages with inconsistent tables: the device flag
    reports that after the fact, it cannot make the pass memory-safe)."""

    class Exceeded(RuntimeError):
        pass

    def __init__(self, caps, device, headroom=1.0625, draws=None):
        self.given_draws = None if draws is None else list(draws)     # the draws of the ordinary pass, to calibrate with
        self.calibrating = caps is None
        self.verifying = False      # replay the capacities but READ every count back and raise before it is used if it does not fit
        self.caps, self.at, self.headroom = ([] if caps is None else list(caps)), 0, float(headroom)
        self.counts = []            # (what, true count) of the calibration pass
        self.consts, self.const_at = [], 0      # device tensors made from host random draws, in program order
        self.overflow = torch.zeros((), dtype=torch.int32, device=device)

    def rewind(self, verifying=False):
        self.at = self.const_at = 0
        self.calibrating, self.verifying = False, verifying
        self.overflow.zero_()

    def draw(self, make, fit, device):
        """A sampler's host-side random draw (CurveFPS phase, VoxelFPS scores, FPS start points).  Calibration: the draw of the
        ordinary pass if one was recorded (``draws``) -- fitted to this pass's sizes by ``fit``: the real points come first in
        every list, so their values carry over and the bounded pass samples exactly what the ordinary pass sampled --, else
        ``make()``; moved to the device once.  Afterwards: that same device tensor (a host -> device copy is not something a
        graph capture can hold, and the draw is a constant of the captured computation anyway)."""
        if self.calibrating:
            t = self.given_draws.pop(0) if self.given_draws else make()
            if fit is not None:
                t = fit(t)
            self.consts.append(t if device is None else t.to(device))
            return self.consts[-1]
        t = self.consts[self.const_at]
        self.const_at += 1
        return t

    def take(self, dev_values, whats):
        out = []
        if self.calibrating:
            for w, v in zip(whats, dev_values.tolist()):
                cap = int(-(-int(v * self.headroom + 32) // 64) * 64)
                self.counts.append((w, int(v)))
                self.caps.append((w, cap))
                out.append(cap)
            self.at = len(self.caps)
            return out
        vals = dev_values.tolist() if self.verifying else None
        for i, w in enumerate(whats):
            if vals is not None and self.at < len(self.caps) and vals[i] > self.caps[self.at][1]:
                raise self.Exceeded("%s: %d exceeds the capacity %d" % (w, vals[i], self.caps[self.at][1]))
            if self.at >= len(self.caps) or self.caps[self.at][0] != w:
                raise RuntimeError("count site %r out of step with the calibration (%s)" % (
                    w, self.caps[self.at][0] if self.at < len(self.caps) else "past the end"))
            cap = int(self.caps[self.at][1])
            self.at += 1
            self.overflow.copy_(torch.maximum(self.overflow, (dev_values[i] > cap).to(torch.int32)))
            out.append(cap)
        return out

    def flag(self, dev_bad):
        """A device-side consistency check (ids sorted ...) that the synchronous path raises on: folded into the flag."""
        if self.calibrating or self.verifying:
            if int(dev_bad.item()) != 0:
                raise AssertionError("batch / curve ids are not sorted (or cloud ids are not 0..B-1)")
            return
        self.overflow.copy_(torch.maximum(self.overflow, (dev_bad != 0).to(torch.int32)))


def bounded():
    return isinstance(counts(), CountBounds)


def draw(make, fit=None, device=None):
    """A host-side random draw of a sampler, through the count resolver when there is one (recorded / replayed / kept as a
    device constant: see CountRecorder.draw, CountBounds.draw)."""
    res = counts()
    if res is None:
        t = make()
        return t if device is None else t.to(device)
    return res.draw(make, fit, device)


def spread_phantoms(pos_q, idx, n_src):
    """Bounded counts only: the slack of a padded sample list names the phantom point (index n_src - 1) over and over, i.e.
    hundreds to thousands of points of the phantom cloud at ONE position -- a worst case for every neighbour search over that
    cloud (one grid cell holds them all: 12 k phantoms cost the A2D2 forward 4 ms).  They are moved onto a lattice of spacing
    16 (no radius of a shipped section exceeds 0.8): isolated points, the cheapest neighbourhoods there are.  The real clouds
    never see the phantom cloud, so nothing of theirs changes."""
    if not bounded():
        return pos_q
    j = torch.arange(idx.numel(), device=idx.device)
    lattice = torch.stack([(j % 64), (j // 64) % 64, j // 4096], dim=1).to(pos_q.dtype) * 16.0
    return torch.where((idx == n_src - 1)[:, None], pos_q + lattice, pos_q)


def _fit_rows(n, fill):
    """fit for draw(): the first entries of the recorded draw, then ``fill`` up to n entries."""
    def fit(t):
        if t.numel() >= n:
            return t[:n].clone()
        return torch.cat([t, torch.full((n - t.numel(),), fill, dtype=t.dtype)])
    return fit


def _count(dev, whats):
    """Host integers for the device-side counts ``dev`` (a 1-D int tensor, one entry per name in ``whats``)."""
    res = counts()
    if isinstance(res, CountBounds):
        return res.take(dev, whats)
    vals = dev.tolist()                                   # host sync, as torch.where / nonzero in the reference
    if res is not None:
        return res.resolve(vals, whats)
    return [int(v) for v in vals]


# --------------------------------------------------------------------------------------
# A1 / A2: segment pointers and curve topology
# --------------------------------------------------------------------------------------

def batch2ptr(batch, with_ends=False):
    """ref src/models/utils/point_ops.py:47-54 (same name, arguments and result)."""
    ids = _i64(batch)
    n = ids.numel()
    dev = ids.device
    starts = torch.empty(n + 1, dtype=torch.int64, device=dev)
    meta = torch.empty(2, dtype=torch.int64, device=dev)
    nbytes = lib().ccn_segment_ptr_workspace_bytes(n)
    ws = workspace(nbytes, dev)
    call("segment_ptr", ptr(ids), n, ptr(starts), None, ptr(meta), ptr(ws), ws.numel())
    runs, bad = (int(v) for v in meta.tolist())           # host sync, as torch.where in the reference
    if bad:
        raise AssertionError("batch2ptr: ids are not sorted")   # reference: assert at point_ops.py:49
    if n == 0:
        return starts[:0] if not with_ends else torch.zeros(2, dtype=torch.int64, device=dev)
    return starts[: runs + 1] if with_ends else starts[1:runs]


class CurveTopology:
    """Curve / cloud CSR tables of one resolution level (built once, shared by every step at
    that level).  ``glob`` is exactly ``curveidx_local2global`` (ref point_ops.py:20-44)."""

    def __init__(self, batch, point2curveidx, num_clouds=None):
        batch, p2c = _i64(batch), _i64(point2curveidx)
        n, dev = batch.numel(), batch.device
        if n == 0:
            raise ValueError("empty point cloud")
        if num_clouds is None:
            num_clouds = int(batch[-1].item()) + 1
        self.n, self.num_clouds, self.batch, self.p2c = n, num_clouds, batch, p2c
        self.glob = torch.empty(n, dtype=torch.int64, device=dev)
        self.cid = torch.empty(n, dtype=torch.int32, device=dev)
        # (bounded counts: the curves beyond the true number are empty runs at the end -- every entry starts as n)
        curve_ptr = (torch.full((n + 1,), n, dtype=torch.int32, device=dev) if bounded()
                     else torch.empty(n + 1, dtype=torch.int32, device=dev))
        self.cloud_ptr = torch.empty(num_clouds + 1, dtype=torch.int64, device=dev)
        meta = torch.empty(4, dtype=torch.int64, device=dev)
        ws = workspace(lib().ccn_curve_topology_workspace_bytes(n, num_clouds), dev)
        call("curve_topology", ptr(batch), ptr(p2c), n, num_clouds, ptr(self.glob), ptr(self.cid), ptr(curve_ptr),
             ptr(self.cloud_ptr), ptr(meta), ptr(ws), ws.numel())
        if bounded():
            counts().flag(meta[1])
            q, longest = _count(meta[0:3:2], ("curves", "longest cloud"))
            q, longest = min(q, n), min(longest, n)
        else:
            q, bad, longest, _ = (int(v) for v in meta.tolist())
            if counts() is not None:
                counts().resolve((q, longest), ("curves", "longest cloud"))
            if bad:
                raise AssertionError("batch / curve ids are not sorted (or cloud ids are not 0..B-1)")
        self.num_curves, self.max_cloud = q, longest
        self.curve_ptr = curve_ptr[: q + 1]
        self.lengths = self.cloud_ptr[1:] - self.cloud_ptr[:-1]
        if bounded():
            # (a cloud longer than the `longest cloud` CAPACITY -- the overflow flag is up -- must not hand the padded (B, Nmax)
            # layouts more points than they hold: the neighbour searches build their grids from these lengths)
            self.lengths = self.lengths.clamp(max=longest)
        if num_clouds == 1:
            self.glob = p2c                                   # quirk Q8: identity for one cloud


def curveidx_local2global(point2curveidx, batch):
    """ref point_ops.py:20-44."""
    return CurveTopology(batch, point2curveidx).glob


# --------------------------------------------------------------------------------------
# generic row gather / scatter (x[idx], x_padded[valid] = x)
# --------------------------------------------------------------------------------------

class GatherRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, index, unique, ascending=False):
        src, index = _mat(src), _i64(index)
        out = _rows(index.numel(), src.size(1), src.device)
        call("gather_rows", ptr(src), _ld(src), ptr(index), index.numel(), src.size(1), ptr(out), _ld(out))
        ctx.save_for_backward(index)
        ctx.rows, ctx.unique, ctx.ascending = src.size(0), unique, ascending
        return out

    @staticmethod
    def backward(ctx, g):
        (index,) = ctx.saved_tensors
        g = _mat(g)
        if ctx.ascending and index.numel():
            dsrc = _rows(ctx.rows, g.size(1), g.device)          # every row written: no memset of the sequence
            call("scatter_rows_fill", ptr(g), _ld(g), ptr(index), index.numel(), g.size(1), ptr(dsrc), _ld(dsrc), ctx.rows, 0)
            return dsrc, None, None, None
        dsrc = _rows(ctx.rows, g.size(1), g.device, zero=True)
        call("scatter_rows", ptr(g), _ld(g), ptr(index), index.numel(), g.size(1), ptr(dsrc), _ld(dsrc),
             0 if ctx.unique else 1)
        return dsrc, None, None, None


def gather_rows(src, index, unique=True, ascending=False):
    """``ascending``: the index is strictly increasing (the curve sequences): the backward pass writes its zero rows itself."""
    return GatherRows.apply(src, index, unique, ascending)


class ScatterRows(torch.autograd.Function):
    """out = zeros(rows, C); out[index] = src  (index entries unique; ``ascending``: strictly increasing, one pass writes
    the zero rows, the halo and the padding columns too)."""

    @staticmethod
    def forward(ctx, src, index, rows, halo=0, ascending=False):
        src, index = _mat(src), _i64(index)
        if ascending and index.numel():
            out = _rows_halo(rows, src.size(1), halo, src.device, init=False) if halo else _rows(rows, src.size(1), src.device)
            buf = out._ccn_halo[0] if halo else out
            call("scatter_rows_fill", ptr(src), _ld(src), ptr(index), index.numel(), src.size(1), ptr(buf), buf.stride(0),
                 buf.size(0), halo)
        else:
            if halo:
                out = _rows_halo(rows, src.size(1), halo, src.device)
                out.zero_()
            else:
                out = _rows(rows, src.size(1), src.device, zero=True)
            call("scatter_rows", ptr(src), _ld(src), ptr(index), index.numel(), src.size(1), ptr(out), _ld(out), 0)
        ctx.save_for_backward(index)
        return out

    @staticmethod
    def backward(ctx, g):
        (index,) = ctx.saved_tensors
        g = _mat(g)
        d = _rows(index.numel(), g.size(1), g.device)
        call("gather_rows", ptr(g), _ld(g), ptr(index), index.numel(), g.size(1), ptr(d), _ld(d))
        return d, None, None, None, None


# --------------------------------------------------------------------------------------
# A3: feature differences fused with the concat
# --------------------------------------------------------------------------------------

class DiffConcat(torch.autograd.Function):
    """cat([x, compute_feature_diffs(x)], dim=1)  (ref fast_conv1d.py:190-205 with :66 / :133)."""

    @staticmethod
    def forward(ctx, x, cid, halo=0):
        x = _mat(x)
        n, c = x.shape
        out = _rows_halo(n, 2 * c, halo, x.device) if halo else _rows(n, 2 * c, x.device)
        call("diff_concat_fwd", ptr(x), _ld(x), ptr(cid), n, c, ptr(out), _ld(out))
        ctx.save_for_backward(x, cid)
        if ACT_TRACE is not None:
            # test hook: the sign of (x[i+1] - x[i]) + (x[i] - x[i-1]) with out-of-curve edges dropped, evaluated with the
            # kernel's own operations in the kernel's order (bit-identical: plain fp32 subtract / add), for the oracle's
            # routed |.| (oracle.torch_ref.routed_abs)
            link = (cid[1:] == cid[:-1])[:, None]
            step = torch.where(link, x[1:] - x[:-1], torch.zeros((), dtype=x.dtype, device=x.device))
            zero = torch.zeros((1, c), dtype=x.dtype, device=x.device)
            sign = torch.sign(torch.cat([step, zero]) + torch.cat([zero, step])).to(torch.int8)
            ACT_TRACE.append((sign if ACT_ROW_MAP is None else sign[ACT_ROW_MAP]).cpu())
        return out

    @staticmethod
    def backward(ctx, g):
        x, cid = ctx.saved_tensors
        g = _mat(g)
        n, c = x.shape
        dx = _rows(n, c, x.device)
        call("diff_concat_bwd", ptr(x), _ld(x), ptr(cid), n, c, ptr(g), _ld(g), ptr(dx), _ld(dx))
        return dx, None, None


def compute_feature_diffs(x, topo):
    return DiffConcat.apply(x, topo.cid)[:, x.size(1):]


# --------------------------------------------------------------------------------------
# A4: shifted-row matrix of the symmetric convolution
# --------------------------------------------------------------------------------------

class Im2Col(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, seg, taps, out16=False):
        # out16 (16-bit storage modes, im2col()): the matrix is written as 16-bit rows for the layer's product and its gradient
        # comes back as bf16 rows -- no fp32 round trip through ccn_cast_rows_h
        x = _mat(x)
        rows, c = x.shape
        ctx.seg, ctx.taps, ctx.c = seg, taps, c
        if out16:
            fdt = _fwd16()
            col = _rows16(rows, taps * c, x.device, fdt)
            call("im2col_fwd_h", ptr(x), _ld(x), ptr(seg), rows, c, taps, ptr(col), _ld(col), 1 if fdt == torch.float16 else 0)
            return col.view(torch.bfloat16) if fdt == torch.float16 else col
        col = _rows(rows, taps * c, x.device)
        call("im2col_fwd", ptr(x), _ld(x), ptr(seg), rows, c, taps, ptr(col), _ld(col))
        return col

    @staticmethod
    def backward(ctx, g):
        rows = g.size(0)
        dx = _rows(rows, ctx.c, g.device)
        if _is_rows16(g):
            call("im2col_bwd_h", ptr(g), _ld(g), ptr(ctx.seg), rows, ctx.c, ctx.taps, ptr(dx), _ld(dx))
        else:
            g = _mat(g.float() if g.dtype != torch.float32 else g)
            call("im2col_bwd", ptr(g), _ld(g), ptr(ctx.seg), rows, ctx.c, ctx.taps, ptr(dx), _ld(dx))
        return dx, None, None, None


def im2col(x, seg, taps):
    """The shifted-row matrix of a curve convolution for ``linear_bn_act``: 16-bit rows in the 16-bit storage modes."""
    out16 = bool(EDGE_OUT16 and _MLP_DTYPE in ("bf16", "fp16") and STORE16 and ACT_TRACE is None and x.size(0) > 0
                 and (taps * x.size(1)) % 8 == 0)
    return _mark16(Im2Col.apply(x, seg, taps, out16), out16)


# --------------------------------------------------------------------------------------
# A16: Linear (+ BatchNorm + activation) layer
# --------------------------------------------------------------------------------------

def _stats_buffer(rows, c, device):
    parts = lib().ccn_stats_rows(rows)
    return torch.empty((parts + 1) * 2 * c, dtype=torch.float64, device=device)


def _aligned_weight(weight):
    """(N, K) weight with rows padded to a multiple of 4 floats (16-byte loads in the GEMM tiles)."""
    k = weight.size(1)
    if k % 4 == 0 and weight.is_contiguous():
        return weight
    wp = _rows(weight.size(0), k, weight.device, zero=True)
    wp.copy_(weight)
    return wp


def _transposed(w, n, k):
    """(k, n) transpose of the (n, >= k) weight ``w`` in an aligned-row buffer, padding columns zero (LDS-tiled kernel: the
    strided torch copy it replaces read the weight uncoalesced, 9 us on average over the 79 layers of the KITTI network)."""
    wt = _rows(k, n, w.device)
    call("transpose_pad", ptr(w), _ld(w), n, k, ptr(wt), _ld(wt))
    return wt


_MLP_DTYPE = os.environ.get("CCN_MLP_DTYPE", "fp32")


def set_mlp_dtype(name):
    """"fp32" (default): exact fp32 MFMA products.  "bf16": the forward and data-gradient products of every
    Linear / conv layer (``LinearBNAct``) round their operands to bf16 inside the GEMM and accumulate in fp32
    (``ccn_gemm_nt_bf16``, BASELINE configs 3 and 5), and so do the weight-gradient products (``ccn_gemm_tn_bf16``);
    weights, activations, gradients and BatchNorm statistics are stored and reduced in fp32.  "bf16x3": fp32-grade
    products on the bf16 matrix cores -- every operand of a forward / data-gradient product is split exactly into three
    bf16 terms and the product assembled from the six leading partial products (``ccn_gemm_nt_x3``: error below one
    fp32 rounding per product, 2.7x fewer matrix-core cycles); weight gradients stay on the fp32 MFMA kernel.  "fp16"
    (BASELINE configs[4], "fp16 features"): the FORWARD products round their operands to fp16 (``ccn_gemm_nt_f16``,
    fp32 accumulation); the data- and weight-gradient products take the bf16 kernels, because gradient magnitudes fall
    below fp16's normal range without loss scaling while bf16 keeps fp32's exponent.  Also settable with the
    environment variable CCN_MLP_DTYPE."""
    global _MLP_DTYPE
    if name not in ("fp32", "bf16", "fp16", "bf16x3"):
        raise ValueError("mlp dtype must be 'fp32', 'bf16', 'fp16' or 'bf16x3'")
    _MLP_DTYPE = name


def mlp_dtype():
    return _MLP_DTYPE


_GEMM_NT = {"fp32": "gemm_nt", "bf16": "gemm_nt_bf16", "fp16": "gemm_nt_f16", "bf16x3": "gemm_nt_x3"}
_GEMM_BWD = {"gemm_nt": "gemm_nt", "gemm_nt_bf16": "gemm_nt_bf16", "gemm_nt_f16": "gemm_nt_bf16", "gemm_nt_x3": "gemm_nt_x3"}
X3_MIN_K = 64            # below this a product is HBM-bound on either kernel


def _nt_name():
    return _GEMM_NT[_MLP_DTYPE]


_NT_SCRATCH = {}
# The paired kernel's tail split (ccn_gemm_nt_ws): OFF by default since round 6.  It regroups the K chains of the tiles it splits;
# measured back to back on the full-width KITTI network, 49 652 points (profiles/r06_parity_split_ab.txt): max-norm distance to the
# fp64 value 8.16e-4 with it against 6.14e-4 without (1.53 x / 1.13 x the CPU oracle's own distance; rms unchanged), for +-0 ms on
# the step (profiles/r05_split_tails.txt: 104.08 vs 104.12 ms).  A change that costs parity margin and buys no time stays an
# option: CCN_NT_SPLIT=1 enables it, the entry points and their tests (tests/test_gpu_gemm_split.py) stay.
NT_SPLIT = os.environ.get("CCN_NT_SPLIT", "0") != "0"
_WS, _WS_ARGS = ("_ws", 2) if NT_SPLIT else ("", 0)       # off: the entries without scratch (what an older library build has)


NT_CAPTURE = None            # graph.py: {} while ONE capture is being recorded -> that capture's scratch buffers by device


def _nt_scratch(device):
    """Scratch of the paired fp32 kernel's tail split (``ccn_gemm_nt_ws`` and its siblings): ONE buffer per (device,
    stream) -- launches on a stream run in order -- whose leading counters start at zero and are left at zero by every
    launch.  -> (pointer, bytes), (None, 0) when the split is switched off.  Under a graph capture the buffer comes from the
    capture's private pool: one per capture when graph.py announced it (``NT_CAPTURE``), else one per call; only the 4 KiB of
    counters are cleared (a 32 MiB fill node per product otherwise)."""
    if not NT_SPLIT:
        return None, 0
    key = (device.index if device.index is not None else torch.cuda.current_device(),
           torch.cuda.current_stream(device).cuda_stream)
    if torch.cuda.is_current_stream_capturing():
        nbytes = int(lib().ccn_gemm_nt_split_workspace_bytes())
        buf = NT_CAPTURE.get(key[0]) if NT_CAPTURE is not None else None
        if buf is None:
            buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
            buf[:4096].zero_()
            if NT_CAPTURE is not None:
                NT_CAPTURE[key[0]] = buf
        return ptr(buf), nbytes
    buf = _NT_SCRATCH.get(key)
    if buf is None:
        nbytes = int(lib().ccn_gemm_nt_split_workspace_bytes())
        buf = _NT_SCRATCH[key] = torch.zeros(nbytes, dtype=torch.uint8, device=device)
    return ptr(buf), buf.numel()


def _gemm_nt(name, x, w, bias, y, m, n, k, stats, xp=None, yp=None):
    """Y = X W^T + b through the entry ``name``; ``xp`` / ``yp`` override the row pointers (a row offset into x / y)."""
    xp = ptr(x) if xp is None else xp
    yp = ptr(y) if yp is None else yp
    if name == "gemm_nt_x3":
        if k >= X3_MIN_K:
            nb = lib().ccn_gemm_x3_workspace_bytes(n, k)
            scratch = workspace(nb, x.device)     # the split weight; stream-ordered reuse by the caching allocator
            call(name, xp, _ld(x), ptr(w), _ld(w), ptr(bias), yp, _ld(y), m, n, k, ptr(stats), ptr(scratch), nb)
            return
        name = "gemm_nt"
    if name == "gemm_nt" and NT_SPLIT:
        call("gemm_nt_ws", xp, _ld(x), ptr(w), _ld(w), ptr(bias), yp, _ld(y), m, n, k, ptr(stats), *_nt_scratch(x.device))
        return
    call(name, xp, _ld(x), ptr(w), _ld(w), ptr(bias), yp, _ld(y), m, n, k, ptr(stats))


_WGRAD_STREAMS = {}
_WGRAD_PENDING = set()
_WGRAD_KEEP = {}                 # device index -> [operand tensors of the products still queued on the side stream]
_WGRAD_KEEP_BYTES = {}
WGRAD_MAX_PENDING_BYTES = 6 << 30


def _wgrad_stream(device):
    """Side stream for weight-gradient products that accumulate straight into a gradient bucket: nothing in the rest of
    the backward pass consumes them, so they run concurrently with the (mostly HBM-bound) BatchNorm / gather passes and
    data-gradient products of the layers below.  ``join_wgrad()`` orders the current stream behind them; it runs as an
    autograd end-of-backward callback, and GradientAllReduce calls it before a bucket is reduced during the pass.
    OFF BY DEFAULT since round 3 (CCN_WGRAD_STREAM=1 / force enables it): measured on the KITTI step it buys 1.4 % (106.8 vs 108.3 ms)
    -- a weight-gradient product released next to a data-gradient product is two MFMA-bound grids taking turns on the CUs, not
    an overlap -- and it costs a third stream, the join protocol below, the GPU_MAX_HW_QUEUES workaround and per-kernel
    durations that are sharing artefacts (profiles/r03_wgrad_stream_ab.txt)."""
    if os.environ.get("CCN_WGRAD_STREAM", "0") == "0" or device.type != "cuda":
        return None
    if (os.environ.get("CCN_WGRAD_STREAM") != "force" and torch.distributed.is_available()
            and torch.distributed.is_initialized()
            and (torch.distributed.get_world_size() > 1 or os.environ.get("CCN_SINGLE_RANK_GROUP"))
            and torch.distributed.get_backend() != "nccl"):
        # Multi-rank runs over gloo (the rehearsal backend: several ranks share ONE GPU) keep the products on the backward
        # stream; over RCCL (one process per GPU) the side stream stays on and the bucket all-reduce is launched from it
        # (parallel.GradientAllReduce._reduce).  In the 2-rank rehearsal (gloo, BOTH ranks on one GPU)
        # the extra stream made a step 3-30x slower, the more hardware queues were in play the worse (no geometry stream:
        # 3x; GPU_MAX_HW_QUEUES=8: no progress) -- queue oversubscription of that one GPU by two processes plus gloo's copy
        # streams.  A single process that owns its GPU and runs the same hooks over a one-rank RCCL group
        # (CCN_SINGLE_RANK_GROUP=nccl) shows no such effect (65.4 vs 65.9 clouds/s with / without the stream: the join
        # before every bucket's all-reduce removes the overlap, so there is nothing to gain either), but a one-rank group
        # launches no RCCL kernels, so the stream stays off where it cannot be validated
        # (CCN_WGRAD_STREAM=force enables it regardless).
        return None
    key = device.index if device.index is not None else torch.cuda.current_device()
    if key not in _WGRAD_STREAMS:
        # (CCN_WGRAD_PRIORITY: the side stream's priority, torch convention -- lower number = served first; the default stream is 0)
        _WGRAD_STREAMS[key] = torch.cuda.Stream(device=device, priority=int(os.environ.get("CCN_WGRAD_PRIORITY", "0")))
    return _WGRAD_STREAMS[key]


def wgrad_stream_of(device):
    """The weight-gradient side stream of ``device`` if products have been queued on it since the last join, else None."""
    key = device.index if device.index is not None else torch.cuda.current_device()
    return _WGRAD_STREAMS.get(key) if key in _WGRAD_PENDING else None


def join_wgrad():
    """Make the current stream wait for every weight-gradient product queued on the side stream, then drop the references
    that kept their operands alive: the memory goes back to the caching allocator only now, when everything the current
    stream does next is ordered behind those products.  (``Tensor.record_stream`` would do the bookkeeping per block,
    but blocks parked that way are not reused in time and the allocator kept growing by ~14 hipMallocs per step.)"""
    for key in list(_WGRAD_PENDING):
        torch.cuda.current_stream(key).wait_stream(_WGRAD_STREAMS[key])
        _WGRAD_PENDING.discard(key)
        _WGRAD_KEEP.pop(key, None)
        _WGRAD_KEEP_BYTES.pop(key, None)


class _WgradScope:
    """``with _WgradScope(into, dy, x):`` runs the enclosed launches on the weight-gradient stream (when the product
    goes into a gradient bucket), after everything queued so far on the current stream."""

    def __init__(self, into, *operands):
        self.ws = _wgrad_stream(into.device) if into is not None else None
        self.operands = operands

    def __enter__(self):
        if self.ws is None:
            return self
        key = self.ws.device.index
        cur = torch.cuda.current_stream()
        if _WGRAD_KEEP_BYTES.get(key, 0) > WGRAD_MAX_PENDING_BYTES:
            join_wgrad()                        # bounds the memory held for products that have not run yet
        self.ws.wait_stream(cur)
        # autograd frees these operands as soon as this backward node returns; they are kept alive until the join
        _WGRAD_KEEP.setdefault(key, []).extend(self.operands)
        _WGRAD_KEEP_BYTES[key] = _WGRAD_KEEP_BYTES.get(key, 0) + sum(t.numel() * t.element_size() for t in self.operands)
        self.scope = torch.cuda.stream(self.ws)
        self.scope.__enter__()
        return self

    def __exit__(self, et, ev, tb):
        if self.ws is not None:
            self.scope.__exit__(et, ev, tb)
            # end of this backward pass: order the backward stream behind the side stream, so that whoever reads
            # .grad after loss.backward() needs no extra call (a stream wait, nothing blocks on the host).  Queued by
            # every scope (the callback is idempotent and costs a microsecond): a flag "already queued" would survive a
            # backward pass that raised after a scope had run, and no later pass would then queue the join.
            torch.autograd.Variable._execution_engine.queue_callback(join_wgrad)
            _WGRAD_PENDING.add(self.ws.device.index)
        return False


def _bucket_view(p):
    """The parameter's gradient-bucket view (parallel.GradientAllReduce) if ``p.grad`` still IS that view.  After an
    optimizer's / module's ``zero_grad(set_to_none=True)`` the attribute is stale (``p.grad`` is None or another tensor):
    the fused path is then off and autograd gets an ordinary gradient tensor, instead of sums piling up in a buffer
    nobody reads or zeroes."""
    if p is None:
        return None
    g = getattr(p, "_ccn_main_grad", None)
    if g is None or not p.is_leaf or not p.requires_grad:
        return None
    if p.grad is None or p.grad.data_ptr() != g.data_ptr():
        return None
    return g


def _has_main_grad(weight):
    return _bucket_view(weight) is not None


def _main_grad(weight, n, k):
    """The bucket view if the weight gradient can be accumulated straight into it: an fp32 (n, k) row-major view.  The
    weight-gradient product accumulates anyway, so writing there saves the zero-fill of a temporary and autograd's add."""
    g = _bucket_view(weight)
    if g is None or g.dtype != torch.float32 or tuple(g.shape) != (n, k) or not g.is_contiguous() or not g.is_cuda:
        return None
    return g


def _main_grad_vec(p, n):
    g = _bucket_view(p)
    if g is None or g.dtype != torch.float32 or tuple(g.shape) != (n,) or not g.is_contiguous() or not g.is_cuda:
        return None
    return g


def _main_grad_note(*params):
    """Forward of a layer that will add these parameters' gradients into their bucket views itself: the all-reduce
    counts the use, and reduces the bucket only after the matching backward (parallel.GradientAllReduce.note_use)."""
    for p in params:
        sync = getattr(p, "_ccn_sync", None) if p is not None else None
        if sync is not None:
            sync.note_use(p)


def _main_grad_cancel(*params):
    """A noted use turned out not to be fused (the bucket view went stale between forward and backward): autograd
    accumulates this gradient and the post-accumulate hook reports it."""
    for p in params:
        sync = getattr(p, "_ccn_sync", None)
        if sync is not None:
            sync.use_cancelled(p)


def _main_grad_done(weight):
    """Tell the gradient all-reduce that this use of the parameter has queued its gradient (autograd is given no tensor
    for it)."""
    sync = getattr(weight, "_ccn_sync", None)
    if sync is not None:
        sync.use_done(weight)
    return None


def _aligned_rows(t):
    """A copy with a 16-byte aligned, multiple-of-4 leading dimension when ``t`` does not have one already (the bf16
    kernel and the LDS-DMA kernels only take such operands; everything the step modules produce already qualifies)."""
    if _ld(t) % 4 == 0 and t.data_ptr() % 16 == 0:
        return t
    out = _rows(t.size(0), t.size(1), t.device)
    out.copy_(t)
    return out


def gemm_tn(dy, x, into=None):
    """dW (N, K) += dY (M, N)^T X (M, K): the weight-gradient product (ref: autograd of F.linear / F.conv1d at
    fast_conv1d.py:183, PyG MLP at base.py:90-125).  ``into``: accumulation target (a gradient-bucket view); a zeroed
    matrix otherwise.  Takes the LDS-DMA kernel when the operands qualify (ccn_gemm_tn_ws), with caller-owned scratch
    for the per-workgroup partial tiles."""
    dy, x = _mat(dy), _mat(x)
    m, n = dy.shape
    k = x.size(1)
    if x.size(0) != m:
        raise ValueError("gemm_tn: dY has %d rows, X has %d" % (m, x.size(0)))
    dw = into if into is not None else _rows(n, k, dy.device, zero=True)
    nb = lib().ccn_gemm_tn_workspace_bytes(m, n, k)
    ws = _tn_scratch(nb, dy.device)
    call("gemm_tn_ws", ptr(dy), _ld(dy), ptr(x), _ld(x), ptr(dw), _ld(dw), m, n, k, ptr(ws), nb)
    return dw


_TN_SCRATCH = {}


def _tn_scratch(nbytes, device):
    """Partial-tile scratch of the weight-gradient kernel: ONE grow-only buffer per (device, stream).  Launches on a
    stream run in order, so every product can reuse it; allocating ~32 MB per call from the caching allocator -- on the
    weight-gradient side stream -- left blocks parked across streams and cost several hipMallocs per step."""
    if nbytes <= 0:
        return None
    key = (device.index if device.index is not None else torch.cuda.current_device(),
           torch.cuda.current_stream(device).cuda_stream)
    buf = _TN_SCRATCH.get(key)
    if buf is None or buf.numel() < nbytes:
        if torch.cuda.is_current_stream_capturing():
            return workspace(nbytes, device)                 # inside a graph capture: the capture's private pool
        buf = _TN_SCRATCH[key] = torch.empty(int(nbytes * 1.25) + 256, dtype=torch.uint8, device=device)
    return buf


def _wgrad(gemm_nt, dy, x, dw, m, n, k):
    """dW += dY^T X for the layer whose forward ran ``gemm_nt`` (bf16 / fp16 modes: bf16 products)."""
    if gemm_nt in ("gemm_nt_bf16", "gemm_nt_f16"):
        call("gemm_tn_bf16", ptr(dy), _ld(dy), ptr(x), _ld(x), ptr(dw), _ld(dw), m, n, k)
        return
    nb = lib().ccn_gemm_tn_workspace_bytes(m, n, k)
    ws = _tn_scratch(nb, dy.device)
    call("gemm_tn_ws", ptr(dy), _ld(dy), ptr(x), _ld(x), ptr(dw), _ld(dw), m, n, k, ptr(ws), nb)


# Gradient sinks: an activation with two consumers whose backward passes both produce a full-size gradient (PointNetConv2's
# message tensor: attend_nn and the softmax aggregation, point_conv.py:89-92).  The consumer that runs first in backward
# registers the gradient tensor it returns (keyed by the activation's address); the Linear layer that consumes the same
# activation then ADDS its data gradient into that tensor (ccn_gemm_nt_acc) and hands autograd nothing, instead of autograd
# summing two E x C tensors in a pass of its own.  Cleared at the end of every backward pass.  CCN_GRAD_SINK=0 disables.
_GRAD_SINK = {}
GRAD_SINK = os.environ.get("CCN_GRAD_SINK", "1") != "0"


def _grad_sink_offer(activation, grad):
    if not GRAD_SINK or not activation.is_cuda:
        return
    torch.autograd.Variable._execution_engine.queue_callback(_GRAD_SINK.clear)     # (idempotent; survives a failed pass)
    _GRAD_SINK[activation.data_ptr()] = (grad, tuple(activation.shape))


def _grad_sink_take(x, m, k):
    hit = _GRAD_SINK.pop(x.data_ptr(), None) if _GRAD_SINK else None
    if hit is None:
        return None
    grad, shape = hit
    if shape != (m, k) or tuple(grad.shape) != (m, k) or grad.stride(1) != 1:
        return None
    return grad


# BatchNorm-backward column sums of a DEFERRED layer taken in the epilogue of its consumer's data-gradient product
# (ccn_gemm_nt_red): the consumer's backward leaves (sums, dZ) here under the address of the deferred output y; the
# producer's backward -- which runs next, a deferred output has exactly one consumer -- takes them when the gradient it is
# handed IS that dZ, and skips its ccn_bn_act_bwd_reduce pass over (dZ, y).  CCN_BN_RED=0 disables.  Cleared after every
# backward pass.
_BN_SUMS = {}
BN_RED = os.environ.get("CCN_BN_RED", "1") != "0"


def _bn_sums_offer(y_prev, sums, dz):
    torch.autograd.Variable._execution_engine.queue_callback(_BN_SUMS.clear)
    _BN_SUMS[y_prev.data_ptr()] = (sums, dz)


def _bn_sums_take(y, g):
    hit = _BN_SUMS.pop(y.data_ptr(), None) if _BN_SUMS else None
    if hit is None:
        return None
    sums, dz = hit
    if dz.data_ptr() != g.data_ptr() or tuple(dz.shape) != tuple(g.shape) or dz.stride(0) != g.stride(0):
        return None
    return sums


class LinearBNAct(torch.autograd.Function):
    """y = act(BN(x W^T + b)) with batch statistics taken in the GEMM epilogue.

    One layer of torch_geometric.nn.MLP as the reference uses it (src/models/base.py:32,64,90-125;
    mlp.py:13) and, with the shifted-row matrix as input, one conv+BN+LeakyReLU layer of
    fast_conv1d.py:71-73 / :140-143.  ``gamma is None`` = plain Linear (MLP's last layer)."""

    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, running_mean, running_var, training, act, eps, momentum, grad_on=True,
                defer=False, xf_par=None, xf_act=0):
        # grad_on: torch.is_grad_enabled() at the call site (inside forward grad mode is always off, and
        # ctx.needs_input_grad ignores no_grad): no backward will come for a pass made under no_grad
        # defer: return the PRE-normalisation product (and the scale / shift table) instead of the activation: the caller
        #   promises that the only consumer is another LinearBNAct, which applies BatchNorm + activation itself
        # xf_par / xf_act: ``x`` is such a deferred product of the previous layer (its 4 x K table and activation code)
        x = _mat(x)
        require_gpu(weight)
        # the second output (the BatchNorm table / an empty placeholder) is not differentiable: without this the engine
        # materialises a zeros tensor for its gradient on every backward call (79 fill launches per KITTI step)
        ctx.set_materialize_grads(False)
        m, k = x.shape
        n = weight.size(0)
        if weight.size(1) != k:
            raise ValueError("linear: input has %d channels, weight expects %d" % (k, weight.size(1)))
        dev = x.device
        w = _aligned_weight(weight.detach())
        gemm_nt = ctx.gemm_nt = _nt_name()
        lazy = (xf_par is not None and gemm_nt == "gemm_nt" and k <= LAZY_ACT_MAX_K and x.data_ptr() % 16 == 0
                and w.data_ptr() % 16 == 0 and bool(lib().ccn_gemm_nt_xf_ok(_ld(x), _ld(w), m, n, k)))
        if xf_par is not None:
            LAZY_ACT_COUNT["fused" if lazy else "written"] += 1
            if LAZY_ACT_LOG is not None:
                LAZY_ACT_LOG.append((m, n, k, lazy))
        if xf_par is not None and not lazy:
            # this product does not take the fused kernel: the previous layer's activation is written after all
            z_in = _rows(m, k, dev)
            call("bn_act_fwd", ptr(x), _ld(x), m, k, ptr(xf_par[0]), ptr(xf_par[1]), int(xf_act), LEAKY_SLOPE, ptr(z_in), _ld(z_in))
            x, xf_par = z_in, None
        ctx.xf_act = int(xf_act) if lazy else None
        y = _rows(m, n, dev)
        has_bn = gamma is not None
        ctx.has_bn, ctx.act, ctx.training, ctx.has_bias = has_bn, ACT[act], bool(training), bias is not None
        none = x.new_empty(0)
        ctx.main_grad_of = weight if (grad_on and ctx.needs_input_grad[1] and _main_grad(weight, n, k) is not None) else None
        ctx.bn_refs = ((gamma, beta) if has_bn and grad_on and ctx.needs_input_grad[3] and ctx.needs_input_grad[4]
                       and _main_grad_vec(gamma, n) is not None and _main_grad_vec(beta, n) is not None else None)
        _main_grad_note(ctx.main_grad_of, *(ctx.bn_refs or ()))
        if gemm_nt != "gemm_nt":
            x = _aligned_rows(x)

        def product(stats):
            if lazy:      # act(BatchNorm(x)) of the previous layer applied between LDS and the matrix cores
                call("gemm_nt_xf" + _WS, ptr(x), _ld(x), ptr(xf_par[0]), ptr(xf_par[1]), ctx.xf_act, LEAKY_SLOPE, ptr(w), _ld(w),
                     ptr(bias), ptr(y), _ld(y), m, n, k, ptr(stats), *_nt_scratch(dev)[:_WS_ARGS])
            else:
                _gemm_nt(gemm_nt, x, w, bias, y, m, n, k, stats)

        if not has_bn:
            product(None)
            ctx.save_for_backward(x, w, xf_par if lazy else none)
            ctx.mark_non_differentiable(none)
            return y, none
        par = torch.empty((4, n), dtype=torch.float32, device=dev)      # scale, shift, mean, rstd
        if training:
            if m < 2:
                raise ValueError("Expected more than 1 value per channel when training")
            stats = _stats_buffer(m, n, dev)
            product(stats)
            call("bn_finalize", ptr(stats), m, n, ptr(gamma), ptr(beta), float(eps), float(momentum),
                 ptr(running_mean), ptr(running_var), ptr(par[0]), ptr(par[1]), ptr(par[2]), ptr(par[3]))
        else:
            product(None)
            call("bn_eval_params", ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var), float(eps), n,
                 ptr(par[0]), ptr(par[1]), ptr(par[2]), ptr(par[3]))
        ctx.save_for_backward(x, w, y, par, xf_par if lazy else none)
        ctx.mark_non_differentiable(par)
        ctx.deferred = bool(defer)
        if defer:
            return y, par
        z = _rows(m, n, dev)
        call("bn_act_fwd", ptr(y), _ld(y), m, n, ptr(par[0]), ptr(par[1]), ctx.act, LEAKY_SLOPE, ptr(z), _ld(z))
        _trace_act(z, ctx.act)
        return z, par

    @staticmethod
    def backward(ctx, g, _gpar=None):
        if g is None:                              # (grads are not materialised: nothing flowed into this layer's output)
            return (None,) * 15
        g = _mat(g)
        dev = g.device
        if ctx.has_bn:
            x, w, y, par, xf_par = ctx.saved_tensors
            m, n = y.shape
            sums = _bn_sums_take(y, g) if ctx.deferred else None     # taken in the consumer's data-gradient epilogue?
            if sums is None:
                sums = _stats_buffer(m, n, dev)
                call("bn_act_bwd_reduce", ptr(g), _ld(g), ptr(y), _ld(y), m, n, ptr(par[0]), ptr(par[1]), ptr(par[2]),
                     ptr(par[3]), ctx.act, LEAKY_SLOPE, ptr(sums))
            dy = _rows(m, n, dev)
            refs = ctx.bn_refs
            gview = _main_grad_vec(refs[0], n) if refs else None
            bview = _main_grad_vec(refs[1], n) if refs else None
            if refs and (gview is None or bview is None):
                _main_grad_cancel(*refs)        # the views went stale since forward: autograd accumulates instead
            if gview is not None and bview is not None:
                # BatchNorm parameter gradients added straight into their gradient-bucket views
                call("bn_act_bwd_apply_ex", ptr(g), _ld(g), ptr(y), _ld(y), m, n, ptr(par[0]), ptr(par[1]), ptr(par[2]),
                     ptr(par[3]), ctx.act, LEAKY_SLOPE, ptr(sums), float(m), 1 if ctx.training else 0, 1, ptr(dy), _ld(dy),
                     ptr(gview), ptr(bview))
                dgamma, dbeta = _main_grad_done(refs[0]), _main_grad_done(refs[1])
            else:
                dgb = torch.empty((2, n), dtype=torch.float32, device=dev)
                call("bn_act_bwd_apply", ptr(g), _ld(g), ptr(y), _ld(y), m, n, ptr(par[0]), ptr(par[1]), ptr(par[2]),
                     ptr(par[3]), ctx.act, LEAKY_SLOPE, ptr(sums), 1 if ctx.training else 0, ptr(dy), _ld(dy),
                     ptr(dgb[0]), ptr(dgb[1]))
                dgamma, dbeta = dgb[0], dgb[1]
        else:
            x, w, xf_par = ctx.saved_tensors
            dy, dgamma, dbeta = g, None, None
            m, n = dy.shape
        k = x.size(1)
        dx = None
        if ctx.needs_input_grad[0]:
            # dX = dY W as an "NT" product with W^T (k x n): both operands then stream along their contiguous
            # index, which is the fastest tile layout (the weight transpose is a few KB..MB)
            wt = _transposed(w, n, k)
            if ctx.gemm_nt != "gemm_nt":
                dy = _aligned_rows(dy)
            sink = _grad_sink_take(x, m, k) if ctx.gemm_nt == "gemm_nt" else None
            if (sink is not None and dy.data_ptr() % 16 == 0 and wt.data_ptr() % 16 == 0
                    and lib().ccn_gemm_nt_acc_ok(_ld(dy), _ld(wt), m, k, n)):
                # the other consumer of x has written its gradient already: add this one to it (autograd gets None)
                call("gemm_nt_acc" + _WS, ptr(dy), _ld(dy), ptr(wt), _ld(wt), ptr(sink), _ld(sink), m, k, n,
                     *_nt_scratch(dev)[:_WS_ARGS])
            elif (BN_RED and ctx.xf_act is not None and ctx.gemm_nt == "gemm_nt" and not (k > 128 and 0 < k % 128 <= 64)
                  and dy.data_ptr() % 16 == 0 and wt.data_ptr() % 16 == 0 and lib().ccn_gemm_nt_acc_ok(_ld(dy), _ld(wt), m, k, n)):
                # x is the previous layer's deferred output y: its BatchNorm-backward column sums come out of this product's
                # epilogue (widths whose plain product is split into a 128-wide and a 64-wide launch keep the separate pass)
                dx = _rows(m, k, dev)
                sums_prev = _stats_buffer(m, k, dev)
                call("gemm_nt_red" + _WS, ptr(dy), _ld(dy), ptr(wt), _ld(wt), ptr(dx), _ld(dx), m, k, n, ptr(x), _ld(x), ptr(xf_par),
                     ctx.xf_act, LEAKY_SLOPE, ptr(sums_prev), *_nt_scratch(dev)[:_WS_ARGS])
                _bn_sums_offer(x, sums_prev, dx)
            else:
                dx = _rows(m, k, dev)
                _gemm_nt(_GEMM_BWD[ctx.gemm_nt], dy, wt, None, dx, m, k, n, None)
        dw = None
        if ctx.needs_input_grad[1]:
            into = _main_grad(ctx.main_grad_of, n, k)
            if into is None and ctx.main_grad_of is not None:
                _main_grad_cancel(ctx.main_grad_of)
            dw = into if into is not None else _rows(n, k, dev, zero=True)
            if ctx.gemm_nt in ("gemm_nt_bf16", "gemm_nt_f16"):
                dy = _aligned_rows(dy)
            xf_act = ctx.xf_act           # (a local: ctx must read the same on a second backward over a retained graph)
            if xf_act is not None and not lib().ccn_gemm_tn_xf_ok(ptr(dy), _ld(dy), ptr(x), _ld(x), m, n, k):
                # (an unaligned incoming gradient: the input activation is written for this product after all)
                z_in = _rows(m, k, dev)
                call("bn_act_fwd", ptr(x), _ld(x), m, k, ptr(xf_par[0]), ptr(xf_par[1]), xf_act, LEAKY_SLOPE, ptr(z_in), _ld(z_in))
                x, xf_act = z_in, None
            with _WgradScope(into, dy, x, xf_par):
                if xf_act is not None:
                    nb = lib().ccn_gemm_tn_workspace_bytes(m, n, k)
                    ws = _tn_scratch(nb, dev)
                    call("gemm_tn_ws_xf", ptr(dy), _ld(dy), ptr(x), _ld(x), ptr(xf_par[0]), ptr(xf_par[1]), xf_act,
                         LEAKY_SLOPE, ptr(dw), _ld(dw), m, n, k, ptr(ws), nb)
                else:
                    _wgrad(ctx.gemm_nt, dy, x, dw, m, n, k)
            if into is not None:
                dw = _main_grad_done(ctx.main_grad_of)
        db = None
        if ctx.has_bias:
            acc = _stats_buffer(m, n, dev)
            db = torch.empty(n, dtype=torch.float32, device=dev)
            call("colsum", ptr(dy), _ld(dy), m, n, ptr(acc), ptr(db))
        return dx, dw, db, dgamma, dbeta, None, None, None, None, None, None, None, None, None, None


# ---- 16-bit STORAGE form of the bf16 MLP mode (BASELINE configs[2]; csrc/ccn_gemm_h.hip) -------------------------------
# In the bf16 mode every product rounds its operands to bf16 anyway.  Here the rounded values are what is STORED: the layer
# input is cast once (or arrives as bf16 rows from the previous layer of the same MLP), the weight is cast per call, the
# BatchNorm-backward gradient dY is written as bf16 rows, and a hidden activation whose only consumer is the next Linear
# is written as bf16 rows (and receives its gradient as bf16 rows).  The products then stream half the bytes through the
# LDS-DMA kernels ccn_gemm_nt_h / ccn_gemm_tn_h.  Forward values are those of the fp32-storage form (one rounding of the
# same fp32 number, earlier); the gradient of a 16-bit hidden activation is rounded to bf16 once more than there
# (oracle.torch_ref mirrors it).  CCN_STORE16=0 keeps the fp32-storage kernels (A/B).
STORE16 = os.environ.get("CCN_STORE16", "1") != "0"
# Round 6: a BatchNorm layer of the 16-bit storage path without its fp32 intermediate -- statistics pass without
# stores (ccn_gemm_nt_h_stats), then product + BatchNorm + activation in one kernel (ccn_gemm_nt_h_bnact): 4 K + 2 N bytes per row
# instead of 2 K + 10 N, the same forward bits; backward reads the layer's output (2 N) instead of the product (4 N) and recovers
# xhat from it (ccn_bn_act_bwd_*_hz).  For layers up to FUSE16_MAX wide / deep: beyond it the product is MFMA-bound and computing
# it twice costs more than its bytes.  CCN_FUSE16=0: the three-kernel form of rounds 3-5 (A/B).
FUSE16 = os.environ.get("CCN_FUSE16", "1") != "0"
FUSE16_MAX = int(os.environ.get("CCN_FUSE16_MAX", "512"))


def _fwd16():
    """dtype of the FORWARD operands in the 16-bit storage form: bf16, or fp16 in the "fp16" mode (whose gradient products
    stay bf16: gradients leave fp16's normal range without loss scaling)."""
    return torch.float16 if _MLP_DTYPE == "fp16" else torch.bfloat16


ROWS16_WATCH = None     # tests: a list collects every 16-bit row matrix allocated (range check of the stored fp16 activations)


def _rows16(rows, cols, device, dtype=torch.bfloat16):
    """(rows, cols) 16-bit matrix whose rows start 16-byte aligned: leading dimension a multiple of 8 elements."""
    ld = (cols + 7) // 8 * 8
    buf = torch.empty((rows, ld), dtype=dtype, device=device)
    out = buf if ld == cols else buf[:, :cols]
    if ROWS16_WATCH is not None:
        ROWS16_WATCH.append(out)
    return out


def _is_rows16(t, dtype=torch.bfloat16):
    return (t.dtype == dtype and t.dim() == 2 and t.stride(1) == 1 and t.stride(0) % 8 == 0
            and t.stride(0) >= t.size(1) and t.data_ptr() % 16 == 0)


F16_XCONV = os.environ.get("CCN_F16_XCONV", "1") != "0"     # (A/B and tests: see LinearBNActH.backward)


def _cast16(x, dtype=torch.bfloat16):
    """fp32 rows -> 16-bit rows (ccn_cast_rows_h); a row matrix of that dtype passes through."""
    if _is_rows16(x, dtype):
        return x
    if x.dtype != torch.float32:
        x = x.float()
    x = _mat(x)
    out = _rows16(x.size(0), x.size(1), x.device, dtype)
    if x.size(0):
        call("cast_rows_h", ptr(x), _ld(x), x.size(0), x.size(1), ptr(out), _ld(out), 1 if dtype == torch.float16 else 0)
    return out


def _bf16_of(x16):
    """bf16 rows of a forward operand: itself in the bf16 mode, bf16(fp16(x)) in the fp16 mode (ccn_f16_to_bf16_rows)."""
    if x16.dtype == torch.bfloat16:
        return x16
    out = _rows16(x16.size(0), x16.size(1), x16.device)
    if x16.size(0):
        call("f16_to_bf16_rows", ptr(x16), _ld(x16), x16.size(0), x16.size(1), ptr(out), _ld(out))
    return out


class LinearBNActH(torch.autograd.Function):
    """LinearBNAct on 16-bit rows (see STORE16 above): y = act(BN(bf16(x) bf16(W)^T + b)), fp32 accumulation and statistics."""

    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, running_mean, running_var, training, act, eps, momentum, grad_on, out16,
                x_f16_bits=False, post=None, post_x=None, dual=False):
        # Plain (no BatchNorm) layers only:
        # post = ("max", grp_ptr, rep_row, n points) / ("attend", offsets, n destinations) with post_x = the messages: the
        #   reduction that follows the layer (CGMax / SegSoftmaxAgg) applied inside this function -- the (rows, N) product never
        #   reaches autograd, so its gradient is written as bf16 rows straight by the reduction's backward (ccn_cg_max_bwd_h,
        #   ccn_seg_softmax_agg_bwd_h) instead of fp32 rows + ccn_cast_rows_h.
        # dual: the product leaves as fp32 rows AND as a 16-bit copy (PointNetConv2's messages: the aggregation reads the
        #   former, attend_nn the latter); backward turns the two gradients into dY rows in one pass (ccn_add_cast_rows_h)
        #   where autograd would add two fp32 tensors and this function would cast the sum.
        # x_f16_bits: ``x`` is a 16-bit activation of the fp16 mode -- fp16 bit patterns in a tensor TYPED bfloat16, so that
        # autograd hands its gradient over as bf16 (an fp16-typed tensor would get an fp16 gradient: out of range)
        require_gpu(x, weight)
        if x_f16_bits:
            x = x.view(torch.float16)
        m, k = x.shape
        n = weight.size(0)
        if weight.size(1) != k:
            raise ValueError("linear: input has %d channels, weight expects %d" % (k, weight.size(1)))
        dev = x.device
        fdt = _fwd16()                                   # forward operand dtype (bf16 / fp16)
        f16 = 1 if fdt == torch.float16 else 0
        ctx.x16_in = bool(x_f16_bits) or (x.dtype == torch.bfloat16 and fdt == torch.bfloat16)
        x16 = _cast16(x, fdt)
        w16 = _cast16(weight.detach(), fdt)
        has_bn = gamma is not None
        # (a 16-bit result whose width is not a multiple of 8 would leave its padding columns unwritten: the three-kernel form)
        # ... and a ReLU layer needs its 16-bit PRE-activation as a second result for the backward pass (xhat of a clipped element
        # cannot be read off z = 0): the kernel writes it in the 16-bit form only)
        fuse = (has_bn and FUSE16 and m > 0 and max(n, k) <= FUSE16_MAX and post is None and not dual
                and not (out16 and n % 8 != 0) and (out16 or ACT[act] != 1))
        y = None if fuse else _rows(m, n, dev)
        ctx.has_bn, ctx.act, ctx.training, ctx.has_bias = has_bn, ACT[act], bool(training), bias is not None
        ctx.out16 = bool(out16 and has_bn)
        none = x16.new_empty(0)          # (unused placeholder kept from the fp32 form)
        ctx.main_grad_of = weight if (grad_on and ctx.needs_input_grad[1] and _main_grad(weight, n, k) is not None) else None
        ctx.bn_refs = ((gamma, beta) if has_bn and grad_on and ctx.needs_input_grad[3] and ctx.needs_input_grad[4]
                       and _main_grad_vec(gamma, n) is not None and _main_grad_vec(beta, n) is not None else None)
        _main_grad_note(ctx.main_grad_of, *(ctx.bn_refs or ()))
        ctx.shape = (m, n, k)

        def product(stats):
            if m:
                call("gemm_nt_h", ptr(x16), _ld(x16), ptr(w16), _ld(w16), ptr(bias), ptr(y), _ld(y), m, n, k, ptr(stats), f16, 0)

        ctx.post, ctx.dual = None, False
        if not has_bn:
            product(None)
            if post is not None and post[0] == "max":
                _, grp_ptr, rep_row, n_pts = post
                out = _rows(n_pts, n, dev)
                arg = torch.empty((n_pts, n), dtype=torch.int32, device=dev)
                call("cg_max_fwd", ptr(y), _ld(y), ptr(grp_ptr), ptr(rep_row), n_pts, n, ptr(out), _ld(out), ptr(arg), work_rows=y.size(0))
                ctx.post = ("max", n_pts)
                ctx.save_for_backward(x16, weight, arg, grp_ptr, rep_row)
                return out
            if post is not None:
                _, offsets, n_dst = post
                msg = _mat(post_x)
                if tuple(msg.shape) != (m, n):
                    raise ValueError("attend: messages %s against scores (%d, %d)" % (tuple(msg.shape), m, n))
                out = _rows(n_dst, n, dev)
                call("seg_softmax_agg_fwd", ptr(msg), _ld(msg), ptr(y), _ld(y), ptr(offsets), n_dst, n, ptr(out), _ld(out), work_rows=msg.size(0))
                ctx.post = ("attend", n_dst)
                ctx.save_for_backward(x16, weight, msg, y, offsets)
                return out
            ctx.save_for_backward(x16, weight)
            if dual:
                ctx.dual = True
                y16 = _cast16(y, fdt)
                return y, (y16.view(torch.bfloat16) if f16 else y16)
            return y
        if post is not None or dual:
            raise ValueError("LinearBNActH: a fused reduction / a dual output follows plain layers only")
        par = torch.empty((4, n), dtype=torch.float32, device=dev)      # scale, shift, mean, rstd
        if training and m < 2:
            raise ValueError("Expected more than 1 value per channel when training")
        # Round 6: the product is computed TWICE instead of stored once (FUSE16, see above): a statistics pass that writes
        # nothing, then product + BatchNorm + activation in one kernel -- the fp32 intermediate y is never written, and backward
        # recovers what it needs from the layer's output z (ccn_bn_act_bwd_*_hz).  Same forward bits as the three-kernel form.
        ctx.from_z = 0
        if fuse:
            if training:
                stats = _stats_buffer(m, n, dev)
                call("gemm_nt_h_stats", ptr(x16), _ld(x16), ptr(w16), _ld(w16), ptr(bias), m, n, k, ptr(stats), f16)
                call("bn_finalize", ptr(stats), m, n, ptr(gamma), ptr(beta), float(eps), float(momentum),
                     ptr(running_mean), ptr(running_var), ptr(par[0]), ptr(par[1]), ptr(par[2]), ptr(par[3]))
            else:
                call("bn_eval_params", ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var), float(eps), n,
                     ptr(par[0]), ptr(par[1]), ptr(par[2]), ptr(par[3]))
            shift = par[1] if bias is None else torch.addcmul(par[1], bias.detach(), par[0])     # (the bias folded into the shift)
            z = _rows16(m, n, dev, fdt) if ctx.out16 else _rows(m, n, dev)
            t16 = _rows16(m, n, dev, fdt) if ctx.act == 1 else None          # (ReLU: the pre-activation, for backward)
            call("gemm_nt_h_bnact", ptr(x16), _ld(x16), ptr(w16), _ld(w16), ptr(par[0]), ptr(shift), ctx.act, LEAKY_SLOPE,
                 ptr(z), _ld(z), ptr(t16), _ld(t16) if t16 is not None else 0, m, n, k, f16, 1 if ctx.out16 else 0)
            ctx.from_z = (2 if f16 else 1) if ctx.out16 else 3
            ctx.z_pre = 1 if t16 is not None else 0
            ctx.save_for_backward(x16, weight, t16 if t16 is not None else z, par)
            if ctx.out16:
                return z.view(torch.bfloat16) if f16 else z
            _trace_act(z, ctx.act)
            return z
        if training:
            stats = _stats_buffer(m, n, dev)
            product(stats)
            call("bn_finalize", ptr(stats), m, n, ptr(gamma), ptr(beta), float(eps), float(momentum),
                 ptr(running_mean), ptr(running_var), ptr(par[0]), ptr(par[1]), ptr(par[2]), ptr(par[3]))
        else:
            product(None)
            call("bn_eval_params", ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var), float(eps), n,
                 ptr(par[0]), ptr(par[1]), ptr(par[2]), ptr(par[3]))
        ctx.save_for_backward(x16, weight, y, par)
        if ctx.out16:
            z = _rows16(m, n, dev, fdt)
            call("bn_act_fwd_h", ptr(y), _ld(y), m, n, ptr(par[0]), ptr(par[1]), ctx.act, LEAKY_SLOPE, ptr(z), _ld(z), f16)
            return z.view(torch.bfloat16) if f16 else z
        z = _rows(m, n, dev)
        call("bn_act_fwd", ptr(y), _ld(y), m, n, ptr(par[0]), ptr(par[1]), ctx.act, LEAKY_SLOPE, ptr(z), _ld(z))
        _trace_act(z, ctx.act)
        return z

    @staticmethod
    def backward(ctx, g, g_copy=None):
        m, n, k = ctx.shape
        dev = ctx.saved_tensors[1].device
        dgamma = dbeta = dpost = None
        if ctx.has_bn:
            x16, weight, y, par = ctx.saved_tensors
            g16 = g.dtype == torch.bfloat16
            if g16 and not _is_rows16(g):
                g, g16 = g.float(), False
            g = g if g16 else _mat(g)
            sums = _stats_buffer(m, n, dev)
            if ctx.from_z:
                call("bn_act_bwd_reduce_hz", ptr(g), 1 if g16 else 0, _ld(g), ptr(y), ctx.from_z, ctx.z_pre, _ld(y), m, n, ptr(par[0]),
                     ptr(par[1]), ptr(par[2]), ptr(par[3]), ctx.act, LEAKY_SLOPE, ptr(sums))
            elif g16:
                call("bn_act_bwd_reduce_h", ptr(g), _ld(g), ptr(y), _ld(y), m, n, ptr(par[0]), ptr(par[1]), ptr(par[2]),
                     ptr(par[3]), ctx.act, LEAKY_SLOPE, ptr(sums))
            else:
                call("bn_act_bwd_reduce", ptr(g), _ld(g), ptr(y), _ld(y), m, n, ptr(par[0]), ptr(par[1]), ptr(par[2]),
                     ptr(par[3]), ctx.act, LEAKY_SLOPE, ptr(sums))
            dy16 = _rows16(m, n, dev)
            refs = ctx.bn_refs
            gview = _main_grad_vec(refs[0], n) if refs else None
            bview = _main_grad_vec(refs[1], n) if refs else None
            if refs and (gview is None or bview is None):
                _main_grad_cancel(*refs)
                gview = bview = None
            fused = gview is not None
            if not fused:
                dgb = torch.empty((2, n), dtype=torch.float32, device=dev)
                gview, bview = dgb[0], dgb[1]
            if ctx.from_z:
                call("bn_act_bwd_apply_hz", ptr(g), 1 if g16 else 0, _ld(g), ptr(y), ctx.from_z, ctx.z_pre, _ld(y), m, n, ptr(par[0]),
                     ptr(par[1]), ptr(par[2]), ptr(par[3]), ctx.act, LEAKY_SLOPE, ptr(sums), float(m), 1 if ctx.training else 0,
                     1 if fused else 0, ptr(dy16), _ld(dy16), ptr(gview), ptr(bview))
            else:
                call("bn_act_bwd_apply_h", ptr(g), 1 if g16 else 0, _ld(g), ptr(y), _ld(y), m, n, ptr(par[0]), ptr(par[1]),
                     ptr(par[2]), ptr(par[3]), ctx.act, LEAKY_SLOPE, ptr(sums), float(m), 1 if ctx.training else 0,
                     1 if fused else 0, ptr(dy16), _ld(dy16), ptr(gview), ptr(bview), 0)
            if fused:
                dgamma, dbeta = _main_grad_done(refs[0]), _main_grad_done(refs[1])
            else:
                dgamma, dbeta = gview, bview
        elif ctx.post is not None and ctx.post[0] == "max":
            x16, weight, arg, grp_ptr, rep_row = ctx.saved_tensors
            sums = par = None
            g = _mat(g.float() if g.dtype != torch.float32 else g)
            dy16 = _rows16(m, n, dev)
            call("cg_max_bwd_h", ptr(g), _ld(g), ptr(arg), ptr(grp_ptr), ptr(rep_row), ctx.post[1], m, n, ptr(dy16), _ld(dy16))
        elif ctx.post is not None:
            x16, weight, msg, att, offsets = ctx.saved_tensors
            sums = par = None
            g = _mat(g.float() if g.dtype != torch.float32 else g)
            dy16, dpost = _rows16(m, n, dev), _rows(m, n, dev)
            call("seg_softmax_agg_bwd_h", ptr(msg), _ld(msg), ptr(att), _ld(att), ptr(offsets), ctx.post[1], n, ptr(g), _ld(g),
                 ptr(dpost), _ld(dpost), ptr(dy16), _ld(dy16), work_rows=msg.size(0))
        else:
            x16, weight = ctx.saved_tensors
            sums = par = None
            if ctx.dual and g is not None and g_copy is not None:
                # gradient of the fp32 rows + gradient of their 16-bit copy -> dY rows, one pass
                ga = _mat(g.float() if g.dtype != torch.float32 else g)
                gb = g_copy if _is_rows16(g_copy) else _cast16(g_copy.float())
                dy16 = _rows16(m, n, dev)
                call("add_cast_rows_h", ptr(ga), _ld(ga), ptr(gb), _ld(gb), m, n, ptr(dy16), _ld(dy16))
                g = None            # (a bias gradient below takes its column sums from dY)
            else:
                if g is None:
                    g = g_copy
                dy16 = _cast16(g)
        dx = None
        if ctx.needs_input_grad[0]:
            wt16 = _rows16(k, n, dev)
            call("transpose_cast_h", ptr(weight), _ld(weight), n, k, ptr(wt16), _ld(wt16), 0)
            if ctx.x16_in:       # the gradient of a 16-bit activation: written as bf16 rows
                dx = _rows16(m, k, dev)
                if m:
                    call("gemm_nt_h", ptr(dy16), _ld(dy16), ptr(wt16), _ld(wt16), None, ptr(dx), _ld(dx), m, k, n, None, 0, 1)
            else:
                dx = _rows(m, k, dev)
                if m:
                    call("gemm_nt_h", ptr(dy16), _ld(dy16), ptr(wt16), _ld(wt16), None, ptr(dx), _ld(dx), m, k, n, None, 0, 0)
        dw = None
        if ctx.needs_input_grad[1]:
            into = _main_grad(ctx.main_grad_of, n, k)
            if into is None and ctx.main_grad_of is not None:
                _main_grad_cancel(ctx.main_grad_of)
            dw = into if into is not None else _rows(n, k, dev, zero=True)
            if m:
                # fp16 mode: the product takes bf16(fp16(x)); the kernel converts the fp16 rows on its MFMA operand
                # (CCN_F16_XCONV=0: a ccn_f16_to_bf16_rows pass first, as before)
                inline = F16_XCONV and x16.dtype == torch.float16
                xb = x16 if inline else _bf16_of(x16)
                with _WgradScope(into, dy16, xb):
                    nb = lib().ccn_gemm_tn_h_workspace_bytes(m, n, k)
                    ws = _tn_scratch(nb, dev)
                    call("gemm_tn_h_xf16" if inline else "gemm_tn_h", ptr(dy16), _ld(dy16), ptr(xb), _ld(xb), ptr(dw), _ld(dw),
                         m, n, k, ptr(ws), nb)
            if into is not None:
                dw = _main_grad_done(ctx.main_grad_of)
        db = None
        if ctx.has_bias:
            if ctx.has_bn:
                # the bias sits in front of the BatchNorm: with batch statistics its gradient is identically zero, with
                # running statistics it is scale * sum(g act'): both from the column sums of the first pass
                db = torch.zeros(n, dtype=torch.float32, device=dev) if ctx.training else par[0] * sums[:n].float()
            else:
                if ctx.post is not None and ctx.post[0] == "attend":
                    # a per-channel shift of the scores leaves every group's softmax unchanged: this gradient is identically
                    # zero (the column sums of datt are rounding noise, 1e-5 of the weight gradient's size in fp32)
                    db = torch.zeros(n, dtype=torch.float32, device=dev)
                else:
                    # (behind a fused max every (point, channel) gradient lands on exactly one row: same column sums; for the
                    # sum of two gradients they are taken from dY itself)
                    src = g if g is not None else dy16[:, :n]
                    gf = _mat(src.float()) if src.dtype != torch.float32 else _mat(src)
                    acc = _stats_buffer(gf.size(0), n, dev)
                    db = torch.empty(n, dtype=torch.float32, device=dev)
                    call("colsum", ptr(gf), _ld(gf), gf.size(0), n, ptr(acc), ptr(db))
        return dx, dw, db, dgamma, dbeta, None, None, None, None, None, None, None, None, None, None, dpost, None


# A hidden MLP layer whose only consumer is the next Linear of the same MLP hands over its PRE-normalisation product; the
# consumer applies BatchNorm + activation to the A fragments inside its GEMM (ccn_gemm_nt_xf, and ccn_gemm_tn_ws_xf for its
# weight gradient), so the activation tensor is never written or read: same bits, one pass over rows x K saved per layer
# (tools/bench_xf.py: 1.34 M x 128 -> 192: 1.20 -> 0.98 ms).  Above K = 256 the pass saved is worth less than the transform
# costs the two products, and the layer is written out as before.  CCN_LAZY_ACT=0 disables.
LAZY_ACT = os.environ.get("CCN_LAZY_ACT", "1") != "0"
LAZY_ACT_MAX_K = int(os.environ.get("CCN_LAZY_ACT_MAX_K", "256"))       # (A/B: 512 and 1024 measured in round 3)
LAZY_ACT_COUNT = {"fused": 0, "written": 0}      # deferred inputs consumed by the fused kernel / written out after all (tests)
LAZY_ACT_LOG = None                               # diagnostics: a list collects (rows, N, K, fused) per deferred input


def apply_post(y, post, post_x=None):
    """The reduction a ``post`` tuple names, as its own autograd function: ("max", grp_ptr, rep_row, n, row_src) = CGMax of
    ``y``; ("attend", offsets, n destinations) = SegSoftmaxAgg of the messages ``post_x`` with scores ``y``."""
    if post[0] == "max":
        return CGMax.apply(y, *post[1:])
    return SegSoftmaxAgg.apply(post_x, y, post[1], post[2], True)


def linear_bn_act(x, weight, bias, bn, training, act, defer=False, post=None, post_x=None, dual=False):
    """bn: a torch.nn.BatchNorm1d used as parameter/buffer container, or None.  ``defer``: the caller feeds the result to
    another linear_bn_act and nothing else (nn.MLP); the result may then be a deferred activation (see LAZY_ACT).
    ``post`` (see apply_post): the result is that reduction of the layer's output; ``dual``: the result is the pair
    (output, its 16-bit copy or None).  Plain layers only; both are fused into the layer's autograd function in the 16-bit
    storage modes (LinearBNActH) and are separate functions / absent otherwise."""
    grad_on = torch.is_grad_enabled()
    if post is not None or dual:
        if bn is not None:
            raise ValueError("linear_bn_act: a reduction / a 16-bit copy follows plain layers only")
        if not (EDGE_OUT16 and _MLP_DTYPE in ("bf16", "fp16") and STORE16 and x.dim() == 2 and x.size(0) > 0
                and MAX_TRACE is None and ACT_TRACE is None and weight.size(0) % 8 == 0):
            y = linear_bn_act(x, weight, bias, None, training, act)
            return (y, None) if dual else apply_post(y, post, post_x)
    if _MLP_DTYPE in ("bf16", "fp16") and STORE16 and x.dim() == 2 and x.size(0) > 0:
        xbits = bool(getattr(x, "_ccn_f16_bits", False))
        if bn is None:
            out = LinearBNActH.apply(x, weight, bias, None, None, None, None, False, None, 0.0, 0.0, grad_on, False, xbits,
                                     post[:4] if post is not None and post[0] == "max" else post, post_x, bool(dual))
            if dual and _MLP_DTYPE == "fp16":
                out[1]._ccn_f16_bits = True
            return out
        if training and bn.track_running_stats:
            bn.num_batches_tracked += 1
        use_batch_stats = training or not bn.track_running_stats
        # (the parity tests read sign tables off the fp32 activation: no 16-bit activation while they are recorded)
        out16 = bool(defer and ACT_TRACE is None and ACT[act] != 0)
        out = LinearBNActH.apply(x, weight, bias, bn.weight, bn.bias, bn.running_mean, bn.running_var, use_batch_stats, act,
                                 bn.eps, bn.momentum if bn.momentum is not None else 0.1, grad_on, out16, xbits)
        if out16 and _MLP_DTYPE == "fp16":
            out._ccn_f16_bits = True
        return out
    if x.dtype in (torch.bfloat16, torch.float16):
        x = x.float()
    pend = getattr(x, "_ccn_deferred", None)
    xf_par, xf_act = pend if pend is not None else (None, 0)
    if bn is None:
        return LinearBNAct.apply(x, weight, bias, None, None, None, None, False, None, 0.0, 0.0, grad_on, False, xf_par,
                                 xf_act)[0]
    if training and bn.track_running_stats:
        bn.num_batches_tracked += 1
    use_batch_stats = training or not bn.track_running_stats
    # (sign tables for the parity tests are taken from the activation itself: no deferral while they are recorded)
    defer = bool(defer and LAZY_ACT and ACT_TRACE is None and _MLP_DTYPE == "fp32" and ACT[act] != 0
                 and weight.size(0) <= LAZY_ACT_MAX_K)
    out, par = LinearBNAct.apply(x, weight, bias, bn.weight, bn.bias, bn.running_mean, bn.running_var, use_batch_stats,
                                 act, bn.eps, bn.momentum if bn.momentum is not None else 0.1, grad_on, defer, xf_par, xf_act)
    if defer:
        out._ccn_deferred = (par, ACT[act])
    return out


class NLLLoss(torch.autograd.Function):
    """mean over the rows with target != ignore_index of -log_softmax(logits)[target] (harness row H; ref
    src/run/kitti_seg.py:184-192): ccn_nll_loss_fwd / _bwd.  ``mean_all``: the KITTI runner's form -- the ignored rows
    contribute zero but stay in the denominator (sum / rows: 0 with a zero gradient for an all-ignored batch, where the
    mean over the counted rows is 0 / 0)."""

    @staticmethod
    def forward(ctx, logits, target, ignore_index=-100, mean_all=False):
        logits, target = _mat(logits), _i64(target)
        rows, c = logits.shape
        if target.numel() != rows:
            raise ValueError("Expected input batch_size (%d) to match target batch_size (%d)." % (rows, target.numel()))
        dev = logits.device
        nb = lib().ccn_nll_loss_blocks(rows)
        lse = torch.empty(rows, dtype=torch.float32, device=dev)
        scratch = torch.empty(2 * nb + 2, dtype=torch.float64, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        call("nll_loss_fwd", ptr(logits), _ld(logits), ptr(target), rows, c, int(ignore_index), ptr(lse), None, ptr(scratch),
             ptr(loss))
        totals = scratch[2 * nb:]                     # (sum of the per-row losses, counted rows)
        if mean_all:
            totals = torch.stack([totals[0], torch.full((), float(rows), dtype=torch.float64, device=dev)])
            loss = (totals[0] / rows).to(torch.float32)
        ctx.save_for_backward(logits, target, lse, totals)
        ctx.ignore = int(ignore_index)
        return loss

    @staticmethod
    def backward(ctx, g):
        logits, target, lse, totals = ctx.saved_tensors
        rows, c = logits.shape
        d = _rows(rows, c, logits.device)
        g = g.to(torch.float32).contiguous()
        call("nll_loss_bwd", ptr(logits), _ld(logits), ptr(target), ptr(lse), rows, c, ctx.ignore, ptr(g),
             ptr(totals), ptr(d), _ld(d))
        return d, None, None, None


# --------------------------------------------------------------------------------------
# A7: CurveFPS
# --------------------------------------------------------------------------------------

def curve_fps(pos, topo, spacing, u):
    """ref src/models/modules/fps_ops.py:16-39; ``u`` = the reference's torch.rand(1) draw."""
    pos = _pos(pos)
    n, dev = pos.size(0), pos.device
    # (bounded counts: the tail of the list names the LAST point -- the phantom point graph.CapturedWholeForward appends as
    # a cloud of its own -- so that every sample past the true count is one more point of that cloud)
    idx = (torch.full((n,), n - 1, dtype=torch.int64, device=dev) if bounded()
           else torch.empty(n, dtype=torch.int64, device=dev))
    count = torch.empty(1, dtype=torch.int64, device=dev)
    ws = workspace(lib().ccn_curve_fps_workspace_bytes(n), dev)
    call("curve_fps", ptr(pos), ptr(topo.cid), ptr(topo.curve_ptr), n, float(spacing), float(u), ptr(idx), ptr(count),
         ptr(ws), ws.numel())
    return idx[: min(_count(count, ("curve-FPS samples",))[0], n)]


# --------------------------------------------------------------------------------------
# A8: radius grouping along curves -> CSR edge list
# --------------------------------------------------------------------------------------

PN_BWD_GATHER = os.environ.get("CCN_PN_BWD_GATHER", "1") != "0"     # A/B: 0 = the round-1..4 backward with fp32 atomics


INV_TORCH = os.environ.get("CCN_INV_TORCH", "0") == "1"


def inverse_lists(src, m):
    """For every one of ``m`` sources the rows r with ``src[r] == source``, ascending: (inv_ptr int32 (m + 1), inv_row int32
    (len(src))).  ``src``: int32 or int64, one-dimensional.  Index-only (ccn_inverse_lists: the library's stable radix sort by source with the
    row number as payload -- no torch.sort / bincount on the geometry stream)."""
    n, dev = src.numel(), src.device
    if INV_TORCH:           # A/B: the round-5 first form (rocprim merge sort + histogram + scan through torch)
        inv_ptr = torch.zeros(m + 1, dtype=torch.int32, device=dev)
        inv_ptr[1:] = torch.cumsum(torch.bincount(src, minlength=m)[:m], 0).to(torch.int32)
        return inv_ptr, torch.sort(src, stable=True)[1].to(torch.int32)
    if n == 0:
        return torch.zeros(m + 1, dtype=torch.int32, device=dev), torch.empty(0, dtype=torch.int32, device=dev)
    inv_ptr = torch.empty(m + 1, dtype=torch.int32, device=dev)
    inv_row = torch.empty(n, dtype=torch.int32, device=dev)
    ws = workspace(lib().ccn_inverse_lists_workspace_bytes(n, m), dev)
    call("inverse_lists", ptr(src), 1 if src.dtype == torch.int64 else 0, n, m, ptr(inv_ptr), ptr(inv_row), ptr(ws), ws.numel())
    return inv_ptr, inv_row


class EdgeList:
    """Edges grouped by destination: ``row`` (destination / query number, non-decreasing),
    ``col`` (source point), ``offsets`` int32 (num_dst + 1)."""

    def __init__(self, row, col, offsets, num_dst, num_src=None):
        self.row, self.col, self.offsets, self.num_dst = row, col, offsets, num_dst
        self.num_edges = row.numel()
        # the inverse of `col` (edges sorted by SOURCE point, ascending edge numbers): what the atomics-free backward of
        # PointNetConv2's first layer gathers through (ccn_pn_edge_bwd_gather).  Index-only, built with the geometry.
        self.inv_src = None
        if (PN_BWD_GATHER and num_src is not None and self.num_edges > 0 and torch.is_grad_enabled() and not bounded()):
            self.inv_src = inverse_lists(col, num_src)


def radius_1d_group_subset(pos, idx, topo, radius):
    """ref point_ops.py:143-193 (quirk Q3 included); returns an EdgeList (row, col as the reference)."""
    pos, idx = _pos(pos), _i64(idx)
    n, m, q, dev = pos.size(0), idx.numel(), topo.num_curves, pos.device
    budget = torch.empty(q + 1, dtype=torch.float32, device=dev)
    offsets = torch.empty(m + 1, dtype=torch.int32, device=dev)
    total = torch.empty(1, dtype=torch.int64, device=dev)
    ws = workspace(lib().ccn_curve_group_subset_workspace_bytes(n, q, m), dev)
    call("curve_group_subset_count", ptr(pos), ptr(topo.cid), ptr(topo.curve_ptr), ptr(topo.p2c), n, q, ptr(idx), m,
         float(radius), ptr(budget), ptr(offsets), ptr(total), ptr(ws), ws.numel())
    e = _count(total, ("curve-group edges",))[0]
    if bounded():
        offsets.clamp_(max=e)           # (capacity exceeded -- the flag is up: every group still ends inside row / col)
    # (bounded counts: edges past the true total belong to no group -- offsets[m] is the true total -- and name point 0)
    row = torch.zeros(e, dtype=torch.int64, device=dev) if bounded() else torch.empty(e, dtype=torch.int64, device=dev)
    col = torch.zeros(e, dtype=torch.int64, device=dev) if bounded() else torch.empty(e, dtype=torch.int64, device=dev)
    call("curve_group_subset_fill_cap", ptr(topo.cid), ptr(topo.curve_ptr), ptr(topo.p2c), n, q, ptr(idx), m, ptr(budget),
         ptr(offsets), ptr(row), ptr(col), e)
    return EdgeList(row, col, offsets, m, num_src=n)


# --------------------------------------------------------------------------------------
# A9: curve interpolation
# --------------------------------------------------------------------------------------

def knn_1d_group_superset_dense(pos, idx, topo, k):
    """ref point_ops.py:196-260 in fixed-width form: (nbr (n,k) int64 -1 padded, weight (n,k))."""
    pos, idx = _pos(pos), _i64(idx)
    n, dev = pos.size(0), pos.device
    nbr = torch.empty((n, k), dtype=torch.int64, device=dev)
    w = torch.empty((n, k), dtype=torch.float32, device=dev)
    ws = workspace(lib().ccn_curve_group_superset_workspace_bytes(n), dev)
    call("curve_group_superset", ptr(pos), ptr(topo.cid), n, ptr(idx), idx.numel(), k, ptr(nbr), ptr(w), ptr(ws),
         ws.numel())
    return nbr, w


def knn_1d_group_superset(pos, idx, topo, k):
    """Same (row, col) edge list as the reference function (host-side flatten of the dense form)."""
    nbr, _ = knn_1d_group_superset_dense(pos, idx, topo, k)
    keep = nbr >= 0
    rows = torch.arange(nbr.size(0), device=nbr.device)[:, None].expand_as(nbr)
    return rows[keep], nbr[keep]


INTERP_GATHER = os.environ.get("CCN_INTERP_GATHER", "1") != "0"


def interp_inverse(nbr, w, m):
    """The (n, k) neighbour table of an interpolation turned around: for every one of the ``m`` coarse rows the list of
    fine rows that read it (sorted), their weights, and every fine row's weight sum.  Built with the geometry (no
    gradient flows through it) so that CurveInterp's backward is an ordered gather instead of k atomic row adds per
    fine row.  None when no gradient is being recorded."""
    if not (INTERP_GATHER and torch.is_grad_enabled()) or m == 0:
        return None
    n, k = nbr.shape
    dev = nbr.device
    inv_ptr = torch.empty(m + 1, dtype=torch.int32, device=dev)
    inv_src = torch.empty(max(n * k, 1), dtype=torch.int32, device=dev)
    inv_w = torch.empty(max(n * k, 1), dtype=torch.float32, device=dev)
    den = torch.empty(max(n, 1), dtype=torch.float32, device=dev)
    ws = workspace(lib().ccn_interp_inverse_workspace_bytes(n, k, m), dev)
    call("interp_inverse", ptr(nbr), ptr(w), n, k, m, ptr(inv_ptr), ptr(inv_src), ptr(inv_w), ptr(den), ptr(ws),
         ws.numel())
    return inv_ptr, inv_src, inv_w, den


class CurveInterp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, nbr, w, inv=None):
        x = _mat(x)
        n, k = nbr.shape
        c = x.size(1)
        y = _rows(n, c, x.device)
        call("interp_fwd", ptr(x), _ld(x), ptr(nbr), ptr(w), n, k, c, ptr(y), _ld(y))
        ctx.save_for_backward(nbr, w, *(inv or ()))
        ctx.m = x.size(0)
        return y

    @staticmethod
    def backward(ctx, g):
        nbr, w, *inv = ctx.saved_tensors
        g = _mat(g)
        n, k = nbr.shape
        c = g.size(1)
        if inv:
            dx = _rows(ctx.m, c, g.device)
            call("interp_bwd_gather", ptr(g), _ld(g), *(ptr(t) for t in inv), ctx.m, c, ptr(dx), _ld(dx))
        else:
            dx = _rows(ctx.m, c, g.device, zero=True)
            call("interp_bwd", ptr(g), _ld(g), ptr(nbr), ptr(w), n, k, c, ptr(dx), _ld(dx))
        return dx, None, None, None


def knn_interpolate_1D(x, idx, pos_y, topo_y, k):
    """ref point_ops.py:344-355."""
    nbr, w = knn_1d_group_superset_dense(pos_y, idx, topo_y, k)
    return CurveInterp.apply(x, nbr, w, interp_inverse(nbr, w, x.size(0)) if x.requires_grad else None)


# --------------------------------------------------------------------------------------
# A11 / A12: fixed-radius kNN and layouts
# --------------------------------------------------------------------------------------

def fast_knn(points1, points2, lengths1, lengths2, K, r, return_dists=False):
    """Drop-in for ref point_ops.py:431-461 (frnn.frnn_grid_points): (B,P1,K) int64, -1 padded."""
    if points1.shape[0] != points2.shape[0]:
        raise ValueError("points1 and points2 must have the same batch  dimension")
    if points1.shape[2] != points2.shape[2]:
        raise ValueError("dimension mismatch: points1 of dimension %d while points2 of dimension %d"
                         % (points1.shape[2], points2.shape[2]))
    if not points1.is_cuda or not points2.is_cuda:
        raise TypeError("for now only cuda version is supported")
    if points1.shape[2] != 3:
        raise ValueError("only 3-D points are supported")
    p1, p2 = _mat(points1), _mat(points2)
    b, n1, n2, dev = p1.size(0), p1.size(1), p2.size(1), p1.device
    if isinstance(r, (float, int)):
        r = torch.full((b,), float(r), dtype=torch.float32, device=dev)    # (on the device at once: no host -> device copy,
    r = r.to(torch.float32)                                                # which a hipGraph capture could not hold)
    if r.numel() == 1:
        r = r.expand(b)
    if r.numel() != b:
        raise ValueError("r must hold one radius or one per cloud")
    r = r.contiguous().to(dev)
    l1, l2 = _i64(lengths1.to(dev)), _i64(lengths2.to(dev))
    idx = torch.empty((b, n1, K), dtype=torch.int64, device=dev)
    d2 = torch.empty((b, n1, K), dtype=torch.float32, device=dev) if return_dists else None
    nbytes = lib().ccn_frnn_grid_bytes(b, n2)
    grid = workspace(nbytes, dev)
    call("frnn_grid_build", ptr(p2), ptr(l2), ptr(r), b, n2, ptr(grid), grid.numel())
    call("frnn_query", ptr(p1), ptr(l1), ptr(r), b, n1, K, ptr(grid), n2, ptr(idx), ptr(d2), None)
    return (idx, d2) if return_dists else idx


def to_batch_padded(t, topo):
    """ref point_ops.py:358-381 using the level's cloud table: (B, Nmax, ...) zero padded + mask."""
    b, nmax = topo.num_clouds, topo.max_cloud
    if b == 1:
        return t.unsqueeze(0), torch.ones((1, t.size(0)), dtype=torch.bool, device=t.device)
    local = torch.arange(topo.n, device=t.device) - topo.cloud_ptr[topo.batch]
    if bounded():
        local = local.clamp(max=nmax - 1)     # (a cloud longer than the capacity: the overflow flag is up, stay in bounds)
    out = torch.zeros((b, nmax) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    out[topo.batch, local] = t
    # (rows i < length of the cloud: the same table as scattering True at every point's slot, without an index_put whose
    # scalar operand is a host -> device copy -- not capturable in a hipGraph)
    mask = torch.arange(nmax, device=t.device)[None, :] < topo.lengths[:, None]
    return out, mask


EXACT_KNN_RADIUS = 1.0e9     # r*r stays finite in float32: every point passes the radius test => exact kNN


def ball_query(points1, points2, lengths1, lengths2, K, radius):
    """pytorch3d.ops.ball_query as the reference calls it (point_ops.py:81): (B,P1,K) int64, first K in index order."""
    p1, p2 = _mat(points1), _mat(points2)
    b, n1, n2, dev = p1.size(0), p1.size(1), p2.size(1), p1.device
    l1, l2 = _i64(lengths1.to(dev)), _i64(lengths2.to(dev))
    idx = torch.empty((b, n1, K), dtype=torch.int64, device=dev)
    d = p1.size(2)
    if p2.size(2) != d:
        raise ValueError("points1 and points2 must have the same dimension")
    if d == 3:
        call("ball_query", ptr(p1), ptr(l1), ptr(p2), ptr(l2), b, n1, n2, K, float(radius), ptr(idx))
    else:       # feature-space search of the dgcnn-rad step
        call("ball_query_nd", ptr(p1), d, ptr(l1), ptr(p2), d, ptr(l2), b, n1, n2, d, K, float(radius), ptr(idx))
    return idx


def frnn_edges(pos_q, topo_q, pos_s, topo_s, k, radius, operation="knn", accel_knn=True):
    """ref point_ops.py:73-111 knn_ball_group_pytorch3d: EdgeList of (query, point) from FRNN (accel_knn), exact
    kNN (accel_knn=False: the same grid search with an unbounded radius) or ball query (K=128, index order)."""
    qp, _ = to_batch_padded(pos_q, topo_q)
    sp, _ = to_batch_padded(pos_s, topo_s)
    if operation == "ball-group":
        assert radius is not None
        k = 128
        nbr = ball_query(qp, sp, topo_q.lengths, topo_s.lengths, k, radius)
    elif accel_knn:
        if radius is None:
            print("Not setting radius for Fast-KNN!")          # quirk Q7
            radius = 0.25
        nbr = fast_knn(qp, sp, topo_q.lengths, topo_s.lengths, k, radius)
    else:
        nbr = fast_knn(qp, sp, topo_q.lengths, topo_s.lengths, k, EXACT_KNN_RADIUS)
    b, p1, dev = nbr.size(0), nbr.size(1), nbr.device
    m = topo_q.n
    counts = (torch.zeros if bounded() else torch.empty)(m + 1, dtype=torch.int32, device=dev)
    call("dense_to_csr_count", ptr(nbr), ptr(topo_q.cloud_ptr), b, p1, k, ptr(counts))
    offsets = torch.empty(m + 1, dtype=torch.int32, device=dev)
    total = torch.empty(1, dtype=torch.int64, device=dev)
    ws = workspace(lib().ccn_exclusive_scan_workspace_bytes(m), dev)
    call("exclusive_scan_i32", ptr(counts), m, ptr(offsets), ptr(total), ptr(ws), ws.numel())
    e = _count(total, ("neighbour-search edges",))[0]
    if bounded():
        offsets.clamp_(max=e)           # (capacity exceeded -- the flag is up: the fill kernel stops at the next offset)
    # (bounded counts: edges past the true total belong to no group and name point 0)
    row = (torch.zeros if bounded() else torch.empty)(e, dtype=torch.int64, device=dev)
    col = (torch.zeros if bounded() else torch.empty)(e, dtype=torch.int64, device=dev)
    call("dense_to_csr_fill", ptr(nbr), ptr(topo_q.cloud_ptr), ptr(topo_s.cloud_ptr), b, p1, k, ptr(offsets), ptr(row),
         ptr(col))
    return EdgeList(row, col, offsets, m, num_src=topo_s.n)


# --------------------------------------------------------------------------------------
# A13: PointNetConv2 message + aggregation
# --------------------------------------------------------------------------------------

class MessageBuild(torch.autograd.Function):
    """cat([x_j, (pos_j - pos_i) / r])  (ref point_conv.py:60-69)."""

    @staticmethod
    def forward(ctx, x_src, pos_src, pos_dst, src, dst, radius):
        c = 0 if x_src is None else x_src.size(1)
        if x_src is not None:
            x_src = _mat(x_src)
        pos_src, pos_dst = _pos(pos_src), _pos(pos_dst)
        e = src.numel()
        msg = _rows(e, c + 3, pos_src.device)
        call("msg_build_fwd", ptr(x_src), _ld(x_src) if c else 0, ptr(pos_src), ptr(pos_dst),
             ptr(src), ptr(dst), e, c, float(radius) if radius is not None else 0.0, ptr(msg), _ld(msg))
        ctx.save_for_backward(src)
        ctx.c, ctx.n_src = c, (0 if x_src is None else x_src.size(0))
        return msg

    @staticmethod
    def backward(ctx, g):
        if ctx.c == 0:
            return None, None, None, None, None, None
        (src,) = ctx.saved_tensors
        g = _mat(g)
        dx = _rows(ctx.n_src, ctx.c, g.device, zero=True)
        call("msg_build_bwd", ptr(g), _ld(g), ptr(src), src.numel(), ctx.c, ptr(dx), _ld(dx))
        return dx, None, None, None, None, None


class EdgeFeat(torch.autograd.Function):
    """cat([x_i, x_j - x_i]) per edge (ref dgcnn.py:227-228, the sparse path's message input)."""

    @staticmethod
    def forward(ctx, x, src, dst, offsets=None, out16=False):
        # offsets (CSR of the edges grouped by destination): backward then adds once per destination and channel on that side
        # (ccn_edge_feat_bwd_csr); out16 (16-bit storage modes, edge_feat()): 16-bit message rows, bf16 gradient back
        x = _mat(x)
        e, c = src.numel(), x.size(1)
        ctx.n, ctx.c = x.size(0), c
        # the CSR backward writes dx[i] for group i = 0 .. num_dst - 1 and assumes group i IS point i of x (x_i = x[dst], dst =
        # the group's index): an edge list over a subset of the points takes the per-edge atomics form instead
        if offsets is not None and offsets.numel() - 1 != x.size(0):
            offsets = None
        ctx.csr = offsets is not None
        ctx.save_for_backward(src, dst, offsets if offsets is not None else src.new_empty(0))
        if out16:
            fdt = _fwd16()
            msg = _rows16(e, 2 * c, x.device, fdt)
            call("edge_feat_fwd_h", ptr(x), _ld(x), ptr(src), ptr(dst), e, c, ptr(msg), _ld(msg), 1 if fdt == torch.float16 else 0)
            return msg.view(torch.bfloat16) if fdt == torch.float16 else msg
        msg = _rows(e, 2 * c, x.device)
        call("edge_feat_fwd", ptr(x), _ld(x), ptr(src), ptr(dst), e, c, ptr(msg), _ld(msg))
        return msg

    @staticmethod
    def backward(ctx, g):
        src, dst, offsets = ctx.saved_tensors
        g16 = _is_rows16(g)
        if not g16:
            g = _mat(g.float() if g.dtype != torch.float32 else g)
        dx = _rows(ctx.n, ctx.c, g.device, zero=True)
        if ctx.csr:
            call("edge_feat_bwd_csr", ptr(g), 1 if g16 else 0, _ld(g), ptr(src), ptr(offsets), offsets.numel() - 1, ctx.n,
                 src.numel(), ctx.c, ptr(dx), _ld(dx))
        else:
            if g16:
                g = _mat(g.float())
            call("edge_feat_bwd", ptr(g), _ld(g), ptr(src), ptr(dst), src.numel(), ctx.c, ptr(dx), _ld(dx))
        return dx, None, None, None, None


def edge_feat(x, edges):
    """cat([x_i, x_j - x_i]) per edge of a CSR edge list (``edges``: row = destination, col = source, offsets) for the message
    MLP: 16-bit rows in the 16-bit storage modes."""
    out16 = bool(EDGE_OUT16 and _MLP_DTYPE in ("bf16", "fp16") and STORE16 and ACT_TRACE is None and edges.num_edges > 0
                 and (2 * x.size(1)) % 8 == 0)
    return _mark16(EdgeFeat.apply(x, edges.col, edges.row, edges.offsets, out16), out16)


class SegSoftmaxAgg(torch.autograd.Function):
    """scatter_add(msg * softmax_per_destination(att))  (ref point_conv.py:89-93)."""

    @staticmethod
    def forward(ctx, msg, att, offsets, num_dst, owns_msg=False):
        # owns_msg: the caller made ``msg`` itself and hands it to exactly two consumers -- the MLP that produced ``att`` and
        # this aggregation (PointNetConv2 / the sparse SGCNN path).  Only then may the gradient sink be used: with a third
        # consumer autograd would accumulate out of place and the in-place add into dmsg would be lost.
        msg, att = _mat(msg), _mat(att)
        c = msg.size(1)
        out = _rows(num_dst, c, msg.device)
        call("seg_softmax_agg_fwd", ptr(msg), _ld(msg), ptr(att), _ld(att), ptr(offsets), num_dst, c, ptr(out), _ld(out), work_rows=msg.size(0))
        _GRAD_SINK.clear()                   # (nothing of an earlier backward pass may survive into this one)
        ctx.save_for_backward(msg, att, offsets)
        ctx.owns_msg = bool(owns_msg)
        return out

    @staticmethod
    def backward(ctx, g):
        msg, att, offsets = ctx.saved_tensors
        g = _mat(g)
        m, c = g.shape
        dmsg, datt = _rows(msg.size(0), c, g.device), _rows(att.size(0), c, g.device)
        call("seg_softmax_agg_bwd", ptr(msg), _ld(msg), ptr(att), _ld(att), ptr(offsets), m, c, ptr(g), _ld(g),
             ptr(dmsg), _ld(dmsg), ptr(datt), _ld(datt), work_rows=msg.size(0))
        if ctx.owns_msg:
            _grad_sink_offer(msg, dmsg)      # attend_nn's first layer adds its data gradient into dmsg (see _GRAD_SINK)
        return dmsg, datt, None, None, None


class SegWSum(torch.autograd.Function):
    """scatter_mean(msg) (mode 0) / scatter_add(msg * sigmoid(att)) (mode 1) over CSR groups (ref point_conv.py:82-88)."""

    @staticmethod
    def forward(ctx, msg, att, offsets, num_dst, mode):
        msg = _mat(msg)
        att = _mat(att) if att is not None else None
        c = msg.size(1)
        out = _rows(num_dst, c, msg.device)
        call("seg_wsum_fwd", ptr(msg), _ld(msg), ptr(att), _ld(att) if att is not None else 0, ptr(offsets), num_dst, c, mode,
             ptr(out), _ld(out), work_rows=msg.size(0))
        ctx.save_for_backward(msg, att if att is not None else msg.new_empty(0), offsets)
        ctx.mode = mode
        return out

    @staticmethod
    def backward(ctx, g):
        msg, att, offsets = ctx.saved_tensors
        g = _mat(g)
        m, c = g.shape
        has_att = ctx.mode == 1
        dmsg = _rows(msg.size(0), c, g.device)
        datt = _rows(msg.size(0), c, g.device) if has_att else None
        call("seg_wsum_bwd", ptr(msg), _ld(msg), ptr(att) if has_att else None, _ld(att) if has_att else 0, ptr(offsets), m, c,
             ctx.mode, ptr(g), _ld(g), ptr(dmsg), _ld(dmsg), ptr(datt), _ld(datt) if has_att else 0, work_rows=msg.size(0))
        return dmsg, datt, None, None, None


class SegMax(torch.autograd.Function):
    """scatter_max over destinations (ref point_conv.py:81-82)."""

    @staticmethod
    def forward(ctx, msg, offsets, num_dst, src=None):
        msg = _mat(msg)
        c = msg.size(1)
        out = _rows(num_dst, c, msg.device)
        arg = torch.empty((num_dst, c), dtype=torch.int32, device=msg.device)
        call("seg_max_fwd", ptr(msg), _ld(msg), ptr(offsets), num_dst, c, ptr(out), _ld(out), ptr(arg), work_rows=msg.size(0))
        ctx.save_for_backward(arg, offsets)
        ctx.e = msg.size(0)
        if MAX_TRACE is not None:
            _trace_max(arg, offsets[:-1], src if src is not None else torch.arange(ctx.e, device=msg.device))
        return out

    @staticmethod
    def backward(ctx, g):
        arg, offsets = ctx.saved_tensors
        g = _mat(g)
        m, c = g.shape
        dmsg = _rows(ctx.e, c, g.device)
        call("seg_max_bwd", ptr(g), _ld(g), ptr(arg), ptr(offsets), m, c, ptr(dmsg), _ld(dmsg), work_rows=ctx.e)
        return dmsg, None, None, None


# --------------------------------------------------------------------------------------
# A15: dense SGCNN gather / masked max
# --------------------------------------------------------------------------------------

class SGGather(torch.autograd.Function):
    """frnn_gather + [f_j, f_self - f_j] over all B*Nmax*(K+1) rows (ref dgcnn.py:166-174)."""

    @staticmethod
    def forward(ctx, x, nbr, cloud_ptr):
        x = _mat(x)
        b, nmax, k = nbr.shape
        c = x.size(1)
        feat = _rows(b * nmax * (k + 1), 2 * c, x.device)
        call("sg_gather_fwd", ptr(x), _ld(x), ptr(nbr), ptr(cloud_ptr), b, nmax, k, c, ptr(feat), _ld(feat))
        ctx.save_for_backward(nbr, cloud_ptr)
        ctx.n, ctx.c = x.size(0), c
        return feat

    @staticmethod
    def backward(ctx, g):
        nbr, cloud_ptr = ctx.saved_tensors
        g = _mat(g)
        b, nmax, k = nbr.shape
        dx = _rows(ctx.n, ctx.c, g.device, zero=True)
        call("sg_gather_bwd", ptr(g), _ld(g), ptr(nbr), ptr(cloud_ptr), b, nmax, k, ctx.c, ptr(dx), _ld(dx))
        return dx, None, None


class SGMax(torch.autograd.Function):
    """masked max over the K+1 slots, packed output rows (ref dgcnn.py:181,187-189,206)."""

    @staticmethod
    def forward(ctx, f, nbr, cloud_ptr, n):
        f = _mat(f)
        b, nmax, k = nbr.shape
        c = f.size(1)
        out = _rows(n, c, f.device)
        arg = torch.empty((n, c), dtype=torch.int32, device=f.device)
        call("sg_max_fwd", ptr(f), _ld(f), ptr(nbr), ptr(cloud_ptr), b, nmax, k, c, ptr(out), _ld(out), ptr(arg))
        ctx.save_for_backward(arg, cloud_ptr)
        ctx.shape = (b, nmax, k, c)
        if MAX_TRACE is not None:       # slot 0 = the point itself, slot s = FRNN entry s-1 (cloud-local index)
            lens = cloud_ptr[1:] - cloud_ptr[:-1]
            cloud = torch.repeat_interleave(torch.arange(b, device=f.device), lens)
            local = torch.arange(n, device=f.device) - cloud_ptr[cloud]
            ids = torch.cat([torch.arange(n, device=f.device)[:, None], nbr[cloud, local] + cloud_ptr[cloud][:, None]], 1)
            _trace_max(arg, torch.arange(n, device=f.device) * (k + 1), ids.reshape(-1))
        return out

    @staticmethod
    def backward(ctx, g):
        arg, cloud_ptr = ctx.saved_tensors
        g = _mat(g)
        b, nmax, k, c = ctx.shape
        df = _rows(b * nmax * (k + 1), c, g.device)
        call("sg_max_bwd", ptr(g), _ld(g), ptr(arg), ptr(cloud_ptr), b, nmax, k, c, ptr(df), _ld(df))
        return df, None, None, None


SG_REDUCE_MODE = {"mean": 0, "weighted-sum": 1, "attend": 2}


class SGReduce(torch.autograd.Function):
    """The mean / weighted-sum / attend reductions of the dense SGCNN path over the K+1 slots (ref dgcnn.py:182-203),
    packed output rows (:206)."""

    @staticmethod
    def forward(ctx, f, att, nbr, cloud_ptr, n, mode):
        f = _mat(f)
        att = _mat(att) if att is not None else None
        b, nmax, k = nbr.shape
        c = f.size(1)
        out = _rows(n, c, f.device)
        call("sg_reduce_fwd", ptr(f), _ld(f), ptr(att), _ld(att) if att is not None else 0, ptr(nbr), ptr(cloud_ptr), b, nmax,
             k, c, mode, ptr(out), _ld(out))
        ctx.save_for_backward(f, att if att is not None else f.new_empty(0), nbr, cloud_ptr)
        ctx.mode = mode
        return out

    @staticmethod
    def backward(ctx, g):
        f, att, nbr, cloud_ptr = ctx.saved_tensors
        g = _mat(g)
        b, nmax, k = nbr.shape
        c = f.size(1)
        has_att = ctx.mode != 0
        df = _rows(f.size(0), c, g.device)
        datt = _rows(f.size(0), c, g.device) if has_att else None
        call("sg_reduce_bwd", ptr(f), _ld(f), ptr(att) if has_att else None, _ld(att) if has_att else 0, ptr(nbr),
             ptr(cloud_ptr), b, nmax, k, c, ctx.mode, ptr(g), _ld(g), ptr(df), _ld(df), ptr(datt), _ld(datt) if has_att else 0)
        return df, datt, None, None, None, None


# --------------------------------------------------------------------------------------
# section 8(f) rows: exact kNN interpolation, voxel and farthest-point sampling
# --------------------------------------------------------------------------------------

KNN_GRID_MIN_POINTS = 4096     # source clouds below this size are searched exhaustively (ccn_knn_points)
KNN_GRID_SCALE = 1.5           # search radius = scale * cbrt(bounding-box volume / points); <= 0 disables the grid path


def knn_points_packed(pos_q, topo_q, pos_s, topo_s, k):
    """pytorch3d.ops.knn_points semantics on packed clouds: (nbr (Nq,k) packed source index, weight)."""
    pos_q, pos_s = _pos(pos_q), _pos(pos_s)
    nq, dev = pos_q.size(0), pos_q.device
    # (bounded counts: the launch covers `longest cloud` CAPACITY queries per cloud; a cloud past it -- the overflow flag is up --
    # leaves rows of the table unwritten, which must still be valid gather indices for the replay to stay inside its buffers)
    alloc = torch.zeros if bounded() else torch.empty
    nbr = alloc((nq, k), dtype=torch.int64, device=dev)
    w = alloc((nq, k), dtype=torch.float32, device=dev)
    b = topo_q.num_clouds
    if topo_s.max_cloud < KNN_GRID_MIN_POINTS or KNN_GRID_SCALE <= 0:
        call("knn_points", ptr(pos_q), ptr(topo_q.cloud_ptr), ptr(pos_s), ptr(topo_s.cloud_ptr), b, topo_q.max_cloud, k,
             ptr(nbr), ptr(w))
        return nbr, w
    # large clouds: hash-grid search at a density-derived radius, exhaustive recomputation of the queries it missed
    radius = torch.empty(b, dtype=torch.float32, device=dev)
    call("knn_cloud_radius", ptr(pos_s), ptr(topo_s.cloud_ptr), b, float(KNN_GRID_SCALE), ptr(radius))
    qp, _ = to_batch_padded(pos_q, topo_q)
    sp, _ = to_batch_padded(pos_s, topo_s)
    idx = fast_knn(qp, sp, topo_q.lengths, topo_s.lengths, k, radius)
    flag = torch.empty(nq + 1, dtype=torch.int32, device=dev)
    call("knn_from_grid", ptr(idx), ptr(pos_q), ptr(topo_q.cloud_ptr), ptr(pos_s), ptr(topo_s.cloud_ptr), b, idx.size(1),
         k, ptr(nbr), ptr(w), ptr(flag))
    offsets = torch.empty(nq + 1, dtype=torch.int32, device=dev)
    total = torch.empty(1, dtype=torch.int64, device=dev)
    ws = workspace(lib().ccn_exclusive_scan_workspace_bytes(nq), dev)
    call("exclusive_scan_i32", ptr(flag), nq, ptr(offsets), ptr(total), ptr(ws), ws.numel())
    missed = torch.empty(nq, dtype=torch.int64, device=dev)
    call("scatter_flagged", ptr(flag), ptr(offsets), nq, ptr(missed))
    call("knn_points_list", ptr(pos_q), ptr(topo_q.cloud_ptr), ptr(pos_s), ptr(topo_s.cloud_ptr), b, k, ptr(missed),
         ptr(total), nq, ptr(nbr), ptr(w))
    knn_points_packed.last_missed = total          # device scalar, for tests / diagnostics
    return nbr, w


def knn_interpolate(x, pos_x, pos_y, topo_x, topo_y, k=3):
    """ref point_ops.py:293-341 knn_interpolate_pytorch3d (inverse squared distance, exact kNN)."""
    nbr, w = knn_points_packed(pos_y, topo_y, pos_x, topo_x, k)
    return CurveInterp.apply(x, nbr, w, interp_inverse(nbr, w, x.size(0)) if x.requires_grad else None)


def voxel_fps(pos, batch, voxel_size, rnd=None):
    """ref src/models/modules/fps_ops.py:42-60 VoxelFPS: one point per occupied voxel, in the
    lexicographic (cloud, voxel) order torch.unique gives.  ``rnd``: the reference's torch.rand(N) draw."""
    pos, batch = _pos(pos), _i64(batch)
    n, dev = pos.size(0), pos.device
    if rnd is None:
        # (the phantom point -- last -- and the slack behind the real points get score 0.5: they never share a voxel with
        # a real point, their scores decide nothing)
        rnd = draw(lambda: torch.rand(n), fit=_fit_rows(n, 0.5), device=dev)
    rnd = rnd.to(device=dev, dtype=torch.float32).contiguous()
    key = torch.empty(n, dtype=torch.int64, device=dev)
    score = torch.empty(n, dtype=torch.float32, device=dev)
    bad = torch.empty(1, dtype=torch.int64, device=dev)
    call("voxel_keys", ptr(pos), ptr(batch), ptr(rnd), n, float(voxel_size), ptr(key), ptr(score), ptr(bad))
    # dense rank of every point's (cloud, voxel) key among the sorted distinct keys = torch.unique(key, return_inverse=True)[1]:
    # ccn_rank_keys (radix sort on the digits in which the keys differ at all)
    meta = torch.empty(2, dtype=torch.int64, device=dev)              # [spread of the keys, number of distinct keys]
    call("key_spread", ptr(key), n, ptr(meta))
    if bounded() and not counts().calibrating and not counts().verifying:
        counts().flag(bad[0])
        digits = 255                    # (no read-back of the key spread: sort on all eight digits)
    else:
        spread, n_bad = int(meta[0].item()), int(bad.item())
        if n_bad:
            raise ValueError("voxel_fps: voxel coordinates exceed the 18-bit key range")
        digits = 255 if bounded() else sum(1 << b for b in range(8) if (spread >> (8 * b)) & 255)
    voxel_of = torch.empty(n, dtype=torch.int64, device=dev)
    nb = lib().ccn_rank_keys_workspace_bytes(n)
    ws = workspace(nb, dev)
    call("rank_keys", ptr(key), n, digits, ptr(voxel_of), ptr(meta[1:]), ptr(ws), nb)
    m = min(_count(meta[1:2], ("voxel samples",))[0], n)
    scratch = torch.empty(m, dtype=torch.int64, device=dev)
    idx = torch.empty(m, dtype=torch.int64, device=dev)
    call("voxel_argmin", ptr(score), ptr(voxel_of), n, m, ptr(scratch), ptr(idx))
    if bounded():
        # voxels past the true number hold no point (index 2^32 - 1 out of the kernel): they name the phantom point, like
        # the tail of every other bounded sample list
        idx = torch.where(idx >= n, torch.full_like(idx, n - 1), idx)
    return idx


_FPS_FALLBACKS = {}


def fps_fallbacks(device):
    """Per device: an int32 counter the library increments once per cloud whose CLUSTER of sampling workgroups gave up (its members
    were not running at the same time) and that the gated one-workgroup launch re-sampled (include/ccn_hip.h: ccn_fps).  The
    samples are the same either way; the counter is there to be looked at (bench.py prints it), never read on the hot path."""
    key = torch.device(device).index or 0
    if key not in _FPS_FALLBACKS:
        _FPS_FALLBACKS[key] = torch.zeros(1, dtype=torch.int32, device=device)
    return _FPS_FALLBACKS[key]


def fps(pos, topo, ratio, start=None):
    """ref point_ops.py:57-70 fps_pytorch3d (sample_farthest_points, random start): sorted packed indices."""
    pos = _pos(pos)
    dev = pos.device
    if bounded():
        # per-cloud sample counts and output offsets stay on the device (the reference's float32 arithmetic, the same
        # torch ops on the other device)
        lengths_d = topo.lengths
        keep = torch.ceil(lengths_d * ratio).long()
        out_ptr_d = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), torch.cumsum(keep, 0)])
        # the random start points: drawn like the synchronous path draws them (torch.randint(length) per cloud) during the
        # calibration pass, the same tensor afterwards (clamped into the cloud: another batch may have shorter clouds)
        start_d = (draw(lambda: torch.tensor([int(torch.randint(max(int(l), 1), (1,))) for l in lengths_d.tolist()],
                                              dtype=torch.int64), device=dev) if start is None else start.to(dev))
        start_d = torch.minimum(start_d, (lengths_d - 1).clamp(min=0))
        total = min(_count(out_ptr_d[-1:], ("FPS samples",))[0], topo.n)
        out_ptr_d = out_ptr_d.clamp(max=total)
        out = torch.full((total,), topo.n - 1, dtype=torch.int64, device=dev)      # (tail: the phantom point)
    else:
        lengths = topo.lengths.cpu()
        keep = torch.ceil(lengths * ratio).long()                        # the reference's float32 arithmetic
        out_ptr = torch.cat([torch.zeros(1, dtype=torch.int64), torch.cumsum(keep, 0)])
        if start is None:
            start = draw(lambda: torch.tensor([int(torch.randint(int(l), (1,))) for l in lengths.tolist()], dtype=torch.int64))
        total = int(out_ptr[-1])
        if counts() is not None:
            counts().resolve((total,), ("FPS samples",))
        out = torch.empty(total, dtype=torch.int64, device=dev)
        start_d, out_ptr_d = start.to(dev), out_ptr.to(dev)     # named: must outlive the asynchronous launch
    nb = lib().ccn_fps_workspace_bytes(topo.n, topo.num_clouds)
    ws = workspace(nb, dev)
    call("fps", ptr(pos), ptr(topo.cloud_ptr), ptr(start_d), ptr(out_ptr_d), topo.num_clouds, topo.max_cloud, topo.n, ptr(ws),
         nb, ptr(fps_fallbacks(dev)), ptr(out))
    # ascending packed indices (the reference sorts them too): ccn_sort_keys on the digits an index < n can differ in
    digits = (1 << max(1, (max(int(topo.n) - 1, 1).bit_length() + 7) // 8)) - 1
    nb = lib().ccn_rank_keys_workspace_bytes(total)
    ws = workspace(nb, dev)
    srt = torch.empty_like(out)
    call("sort_keys", ptr(out), total, digits, ptr(srt), ptr(ws), nb)
    return srt


class SGEdgeLayer(torch.autograd.Function):
    """First edge layer of the dense SGCNN path in algebraic form (ref dgcnn.py:166-177 + the first
    Linear/BatchNorm/activation of ``self.nn``):  W [x_j ; x_i - x_j] = (Wa - Wb) x_j + Wb x_i.

    ``ps`` (N, 2*Co) holds P = X (Wa-Wb)^T and S = X Wb^T per point; the dense row (b, i, slot) is
    P[neighbour] + S[i].  Batch statistics are taken over all B*Nmax*(K+1) rows (quirk Q4) without
    materialising the pre-activation tensor; backward recomputes it from ``ps``."""

    @staticmethod
    def forward(ctx, ps, nbr, cloud_ptr, gamma, beta, running_mean, running_var, training, act, eps, momentum):
        ps = _mat(ps)
        b, nmax, k = nbr.shape
        co = ps.size(1) // 2
        rows = b * nmax * (k + 1)
        dev = ps.device
        has_bn = gamma is not None
        ctx.has_bn, ctx.act, ctx.training = has_bn, ACT[act], bool(training)
        par = None
        if has_bn:
            par = torch.empty((4, co), dtype=torch.float32, device=dev)
            if training:
                nparts = lib().ccn_sg_edge_stats_rows(b, nmax, co)
                partial = torch.empty((nparts + 1) * 2 * co, dtype=torch.float64, device=dev)
                call("sg_edge_stats", ptr(ps), _ld(ps), None, ptr(nbr), ptr(cloud_ptr), b, nmax, k, co, ptr(partial))
                call("bn_finalize_n", ptr(partial), nparts, rows, co, ptr(gamma), ptr(beta), float(eps),
                     float(momentum), ptr(running_mean), ptr(running_var), ptr(par[0]), ptr(par[1]), ptr(par[2]),
                     ptr(par[3]))
            else:
                call("bn_eval_params", ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var), float(eps), co,
                     ptr(par[0]), ptr(par[1]), ptr(par[2]), ptr(par[3]))
        z = _rows(rows, co, dev)
        call("sg_edge_apply", ptr(ps), _ld(ps), None, ptr(nbr), ptr(cloud_ptr), b, nmax, k, co,
             ptr(par[0]) if has_bn else None, ptr(par[1]) if has_bn else None, ctx.act, LEAKY_SLOPE, ptr(z), _ld(z))
        ctx.save_for_backward(ps, nbr, cloud_ptr, par if has_bn else ps.new_empty(0))
        _trace_act(z, ctx.act)
        return z

    @staticmethod
    def backward(ctx, g):
        ps, nbr, cloud_ptr, par = ctx.saved_tensors
        g = _mat(g)
        b, nmax, k = nbr.shape
        co = ps.size(1) // 2
        dev = g.device
        sums = dgamma = dbeta = None
        pp = [None] * 4
        if ctx.has_bn:
            pp = [ptr(par[0]), ptr(par[1]), ptr(par[2]), ptr(par[3])]
            nparts = lib().ccn_sg_edge_stats_rows(b, nmax, co)
            partial = torch.empty((nparts + 1) * 2 * co, dtype=torch.float64, device=dev)
            call("sg_edge_bwd_stats", ptr(ps), _ld(ps), None, ptr(nbr), ptr(cloud_ptr), b, nmax, k, co, ptr(g), _ld(g),
                 *pp, ctx.act, LEAKY_SLOPE, ptr(partial))
            sums = partial[nparts * 2 * co:]
            call("reduce_partials", ptr(partial), nparts, 2 * co, ptr(sums))
            dgb = sums[:2 * co].float()                   # one conversion, two views
            dbeta, dgamma = dgb[:co], dgb[co:]
        dps = _rows(ps.size(0), 2 * co, dev, zero=True)
        call("sg_edge_bwd", ptr(ps), _ld(ps), None, ptr(nbr), ptr(cloud_ptr), b, nmax, k, co, ptr(g), _ld(g), *pp,
             ctx.act, LEAKY_SLOPE, ptr(sums) if sums is not None else None, 1 if ctx.training else 0, ptr(dps), _ld(dps))
        return dps, None, None, dgamma, dbeta, None, None, None, None, None, None


def sg_edge_layer(ps, nbr, cloud_ptr, bn, training, act):
    if bn is None:
        return SGEdgeLayer.apply(ps, nbr, cloud_ptr, None, None, None, None, False, None, 0.0, 0.0)
    if training and bn.track_running_stats:
        bn.num_batches_tracked += 1
    use_batch_stats = training or not bn.track_running_stats
    return SGEdgeLayer.apply(ps, nbr, cloud_ptr, bn.weight, bn.bias, bn.running_mean, bn.running_var, use_batch_stats,
                             act, bn.eps, bn.momentum if bn.momentum is not None else 0.1)


class PNEdgeLayer(torch.autograd.Function):
    """First layer of PointNetConv2's ``local_nn`` in algebraic form (ref point_conv.py:35-93):
    ``W [x_j ; (p_j - p_i)/r] + b = PX[j] + Wp (p_j - p_i)/r + b`` with ``PX = X Wx^T`` computed once per SOURCE POINT
    (the caller's GEMM) instead of once per edge; BatchNorm statistics over the E edges and the activation are fused
    into the gather passes.  Exact up to fp32 re-association of the C+3 term dot product."""

    @staticmethod
    def forward(ctx, px, wp, bias, pos_src, pos_dst, src, dst, radius, gamma, beta, running_mean, running_var, training,
                act, eps, momentum, out16=False, inv=None):
        ctx.inv = inv
        px, wp = _mat(px), _mat(wp.contiguous())
        e, co, dev = src.numel(), px.size(1), px.device
        has_bn = gamma is not None
        ctx.has_bn, ctx.act, ctx.training, ctx.radius = has_bn, ACT[act], bool(training), float(radius or 0.0)
        ctx.has_bias = bias is not None
        geo = (ptr(pos_src), ptr(pos_dst), ptr(src), ptr(dst), e, co, ctx.radius)
        par = None
        if has_bn:
            par = torch.empty((4, co), dtype=torch.float32, device=dev)
            if training:
                nparts = lib().ccn_pn_edge_stats_rows(e, co)
                partial = torch.empty((nparts + 1) * 2 * co, dtype=torch.float64, device=dev)
                call("pn_edge_stats", ptr(px), _ld(px), ptr(wp), _ld(wp), ptr(bias), *geo, ptr(partial))
                call("bn_finalize_n", ptr(partial), nparts, e, co, ptr(gamma), ptr(beta), float(eps), float(momentum),
                     ptr(running_mean), ptr(running_var), ptr(par[0]), ptr(par[1]), ptr(par[2]), ptr(par[3]))
            else:
                call("bn_eval_params", ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var), float(eps), co,
                     ptr(par[0]), ptr(par[1]), ptr(par[2]), ptr(par[3]))
        ctx.save_for_backward(px, wp, bias if bias is not None else px.new_empty(0), pos_src, pos_dst, src, dst,
                              par if has_bn else px.new_empty(0))
        if out16:       # 16-bit storage modes: the next Linear of the MLP reads 16-bit rows (edge_out16)
            fdt = _fwd16()
            z = _rows16(e, co, dev, fdt)
            call("pn_edge_apply_h", ptr(px), _ld(px), ptr(wp), _ld(wp), ptr(bias), *geo, ptr(par[0]) if has_bn else None,
                 ptr(par[1]) if has_bn else None, ctx.act, LEAKY_SLOPE, ptr(z), _ld(z), 1 if fdt == torch.float16 else 0)
            return z.view(torch.bfloat16) if fdt == torch.float16 else z
        z = _rows(e, co, dev)
        call("pn_edge_apply", ptr(px), _ld(px), ptr(wp), _ld(wp), ptr(bias), *geo, ptr(par[0]) if has_bn else None,
             ptr(par[1]) if has_bn else None, ctx.act, LEAKY_SLOPE, ptr(z), _ld(z))
        _trace_act(z, ctx.act)
        return z

    @staticmethod
    def backward(ctx, g):
        px, wp, bias, pos_src, pos_dst, src, dst, par = ctx.saved_tensors
        h = "_h" if _is_rows16(g) else ""          # the gradient of a 16-bit activation arrives as bf16 rows
        g = g if h else _mat(g.float() if g.dtype != torch.float32 else g)
        e, co, dev = src.numel(), px.size(1), g.device
        bias_p = ptr(bias) if ctx.has_bias else None
        geo = (ptr(pos_src), ptr(pos_dst), ptr(src), ptr(dst), e, co, ctx.radius)
        sums = dgamma = dbeta = None
        pp = [None] * 4
        if ctx.inv is not None:
            # round 5: no atomics, one pass over dZ per index order (ccn_pn_edge_bwd_sums / _gather / _finish)
            inv_ptr, inv_edge = ctx.inv
            nsrc = px.size(0)
            if ctx.has_bn:
                tab = par
            else:
                tab = torch.zeros((4, co), dtype=torch.float32, device=dev)
                tab[0].fill_(1.0)
            pp = [ptr(tab[0]), ptr(tab[1]), ptr(tab[2]), ptr(tab[3])]
            dz16 = 1 if h else 0
            nparts = lib().ccn_pn_edge_stats_rows(e, co)
            width = 9 * co + 4
            partial = torch.empty((nparts + 1) * width, dtype=torch.float64, device=dev)
            call("pn_edge_bwd_sums", ptr(px), _ld(px), ptr(wp), _ld(wp), bias_p, *geo, ptr(g), dz16, _ld(g), *pp, ctx.act,
                 LEAKY_SLOPE, ptr(partial))
            pq = _rows(nsrc, 2 * co, dev)
            call("pn_edge_bwd_gather", ptr(px), _ld(px), ptr(wp), _ld(wp), bias_p, ptr(pos_src), ptr(pos_dst), ptr(dst),
                 ptr(inv_ptr), ptr(inv_edge), nsrc, co, ctx.radius, ptr(g), dz16, _ld(g), *pp, ctx.act, LEAKY_SLOPE, ptr(pq),
                 _ld(pq), work_rows=e)
            sums = partial[nparts * width:]
            call("reduce_partials", ptr(partial), nparts, width, ptr(sums))
            if ctx.has_bn:
                dgb = sums[:2 * co].float()
                dbeta, dgamma = dgb[:co], dgb[co:]
            dpx = _rows(nsrc, co, dev)
            dw4 = torch.empty((4, co), dtype=torch.float32, device=dev)
            call("pn_edge_bwd_finish", ptr(pq), _ld(pq), ptr(inv_ptr), nsrc, e, co, pp[0], ptr(sums),
                 1 if (ctx.training and ctx.has_bn) else 0, ptr(dpx), _ld(dpx), ptr(dw4))
            dwp = dw4[:3].t().contiguous()
            dbias = dw4[3].contiguous() if ctx.has_bias else None
            return (dpx, dwp, dbias) + (None,) * 5 + (dgamma, dbeta) + (None,) * 8
        if ctx.has_bn:
            pp = [ptr(par[0]), ptr(par[1]), ptr(par[2]), ptr(par[3])]
            nparts = lib().ccn_pn_edge_stats_rows(e, co)
            partial = torch.empty((nparts + 1) * 2 * co, dtype=torch.float64, device=dev)
            call("pn_edge_bwd_stats" + h, ptr(px), _ld(px), ptr(wp), _ld(wp), bias_p, *geo, ptr(g), _ld(g), *pp, ctx.act,
                 LEAKY_SLOPE, ptr(partial))
            sums = partial[nparts * 2 * co:]
            call("reduce_partials", ptr(partial), nparts, 2 * co, ptr(sums))
            dgb = sums[:2 * co].float()                   # one conversion, two views
            dbeta, dgamma = dgb[:co], dgb[co:]
        dpx = _rows(px.size(0), co, dev, zero=True)
        nw = lib().ccn_pn_edge_bwd_rows(e)
        wpart = torch.empty((nw + 1) * 4 * co, dtype=torch.float64, device=dev)       # every partial row is written
        call("pn_edge_bwd" + h, ptr(px), _ld(px), ptr(wp), _ld(wp), bias_p, *geo, ptr(g), _ld(g), *pp, ctx.act, LEAKY_SLOPE,
             ptr(sums) if sums is not None else None, 1 if (ctx.training and ctx.has_bn) else 0, ptr(dpx), _ld(dpx),
             ptr(wpart))
        tot = wpart[nw * 4 * co:]
        call("reduce_partials", ptr(wpart), nw, 4 * co, ptr(tot))
        tot = tot.view(4, co).float()
        dwp = tot[:3].t().contiguous()
        dbias = tot[3].contiguous() if ctx.has_bias else None
        return (dpx, dwp, dbias) + (None,) * 5 + (dgamma, dbeta) + (None,) * 8


EDGE_OUT16 = os.environ.get("CCN_EDGE_OUT16", "1") != "0"      # (A/B and tests: 0 = fp32 rows + ccn_cast_rows_h as before)


def edge_out16(mlp, channels):
    """May the algebraic first layer of ``mlp`` (an nn.MLP) write its activation as 16-bit rows?  Yes in the 16-bit storage
    modes when that activation is a hidden one whose only consumer is the next Linear of the same MLP -- the rows the
    consumer would otherwise get through ccn_cast_rows_h, with the same rounding (and a bf16 gradient back, as for every
    other hidden activation of those modes)."""
    return bool(EDGE_OUT16 and _MLP_DTYPE in ("bf16", "fp16") and STORE16 and ACT_TRACE is None and len(mlp.norms) > 0
                and (len(mlp.norms) > 1 or mlp.plain_last) and mlp.dropout == 0.0 and ACT[mlp.act] != 0
                and channels % 8 == 0)


def _mark16(z, out16):
    if out16 and _MLP_DTYPE == "fp16":
        z._ccn_f16_bits = True         # fp16 bit patterns in a bfloat16-typed tensor (see LinearBNActH)
    return z


def pn_edge_layer(px, wp, bias, pos_src, pos_dst, edges, radius, bn, training, act, out16=False):
    pos_src, pos_dst = _pos(pos_src), _pos(pos_dst)
    inv = getattr(edges, "inv_src", None)
    if inv is not None and inv[0].numel() != px.size(0) + 1:
        inv = None                      # (an edge list whose sources are not the rows of px: the atomic form)
    if bn is None:
        return PNEdgeLayer.apply(px, wp, bias, pos_src, pos_dst, edges.col, edges.row, radius, None, None, None, None,
                                 False, None, 0.0, 0.0, False, inv)
    if training and bn.track_running_stats:
        bn.num_batches_tracked += 1
    use_batch_stats = training or not bn.track_running_stats
    out16 = bool(out16 and edges.num_edges > 0)
    return _mark16(PNEdgeLayer.apply(px, wp, bias, pos_src, pos_dst, edges.col, edges.row, radius, bn.weight, bn.bias,
                                     bn.running_mean, bn.running_var, use_batch_stats, act, bn.eps,
                                     bn.momentum if bn.momentum is not None else 0.1, out16, inv), out16)


# --------------------------------------------------------------------------------------
# A15 on compact rows: the dense SGCNN computation without its duplicate rows (see ccn_hip.h, "COMPACT rows")
# --------------------------------------------------------------------------------------

CG_BWD_GATHER = os.environ.get("CCN_CG_BWD_GATHER", "1") != "0"     # A/B: 0 = the round-1..4 backward with fp32 atomics


class SGCompact:
    """Row structure of one SGCNN call: real rows grouped by point, one weighted representative per point with empty
    FRNN slots, one weighted row for all padding rows.  Index-only: built inside a geometry block."""

    def __init__(self, nbr, topo):
        b, nmax, k = nbr.shape
        n, dev = topo.n, nbr.device
        # (bounded counts: the launches cover `longest cloud` CAPACITY points per cloud; the points of a cloud past it -- the overflow
        # flag is up -- must read as "no rows, no representative", not as whatever the allocation held: found by replaying a
        # deliberately overflowing batch, tools/dbg_overflow.py)
        alloc = torch.zeros if bounded() else torch.empty
        cnt = alloc(n + 1, dtype=torch.int32, device=dev)
        has = alloc(n + 1, dtype=torch.int32, device=dev)
        call("cg_count", ptr(nbr), ptr(topo.cloud_ptr), b, nmax, k, ptr(cnt), ptr(has))
        self.grp_ptr = torch.empty(n + 1, dtype=torch.int32, device=dev)
        rep_off = torch.empty(n + 1, dtype=torch.int32, device=dev)
        totals = torch.empty(2, dtype=torch.int64, device=dev)
        ws = workspace(lib().ccn_exclusive_scan_workspace_bytes(n), dev)
        call("exclusive_scan_i32", ptr(cnt), n, ptr(self.grp_ptr), ptr(totals[0:1]), ptr(ws), ws.numel())
        call("exclusive_scan_i32", ptr(has), n, ptr(rep_off), ptr(totals[1:2]), ptr(ws), ws.numel())
        if bounded():
            # real rows [0, e_cap) (the true total is grp_ptr[n]; rows past it belong to no group and name point 0),
            # representatives [e_cap, e_cap + n) -- any point may have one --, the padding row behind them
            e, ne = _count(totals, ("compact SGCNN rows", "compact SGCNN representatives"))[0], n
            self.grp_ptr.clamp_(max=e)  # (capacity exceeded -- the flag is up: every group still ends inside row_src / the rows)
        else:
            e, ne = _count(totals, ("compact SGCNN rows", "compact SGCNN representatives"))
        self.n, self.k, self.e, self.ne = n, k, e, ne
        self.rows = e + ne + 1
        self.count = float(b * nmax * (k + 1))                    # rows of the dense layout = sum of all weights
        self.row_src = (torch.zeros(e, dtype=torch.int32, device=dev) if bounded()
                        else torch.empty(e, dtype=torch.int32, device=dev))
        self.rep_row = (torch.full((n,), -1, dtype=torch.int32, device=dev) if bounded()
                        else torch.empty(n, dtype=torch.int32, device=dev))
        self.row_w = alloc(ne + 1, dtype=torch.float32, device=dev)
        call("cg_fill", ptr(nbr), ptr(topo.cloud_ptr), b, nmax, k, ptr(self.grp_ptr), ptr(rep_off), e, ptr(self.row_src),
             ptr(self.rep_row), ptr(self.row_w))
        self.row_w[ne:].fill_(float((b * nmax - n) * (k + 1)))    # the padding row stands for all padding rows
        # the inverse of row_src (rows sorted by SOURCE point, ascending row numbers) + the owner of every row: what the
        # atomics-free backward of the first layer gathers through (ccn_cg_edge_bwd_gather).  Index-only, built with the
        # geometry; not needed when no gradient is being recorded.
        self.inv = None
        if CG_BWD_GATHER and torch.is_grad_enabled() and not bounded():
            inv_ptr, order = inverse_lists(self.row_src, n)
            row_dst = torch.empty(e, dtype=torch.int32, device=dev)
            call("group_owner", ptr(self.grp_ptr), n, e, ptr(row_dst))
            self.inv = (inv_ptr, order, row_dst)

    def tensors(self):
        return (self.grp_ptr, self.row_src, self.rep_row, self.row_w)

    def dense_row_map(self, nbr, topo):
        """Compact row of every row (b, i, slot) of the reference's dense B*Nmax*(K+1)-row layout (test hook: sign tables
        of the compact computation in the oracle's layout)."""
        b, nmax, k = nbr.shape
        dev = nbr.device
        live = torch.arange(nmax, device=dev)[None, :] < topo.lengths[:, None]                  # (B, Nmax)
        p = (topo.cloud_ptr[:-1, None] + torch.arange(nmax, device=dev)[None, :]).clamp(max=self.n - 1)
        valid = torch.cat([torch.ones((b, nmax, 1), dtype=torch.bool, device=dev), nbr >= 0], dim=2)
        rank = torch.cumsum(valid.long(), dim=2) - 1
        real = self.grp_ptr.long()[p][:, :, None] + rank
        rep = self.rep_row.long()[p][:, :, None].expand(-1, -1, k + 1)
        idx = torch.where(valid, real, rep)
        idx = torch.where(live[:, :, None], idx, torch.full_like(idx, self.e + self.ne))
        return idx.reshape(-1)


class CGEdgeLayer(torch.autograd.Function):
    """``SGEdgeLayer`` on compact rows: same values for the real rows, BatchNorm statistics identical to the dense
    B*Nmax*(K+1)-row computation (weighted representatives), output (E + Ne + 1, Co)."""

    @staticmethod
    def forward(ctx, ps, grp_ptr, row_src, rep_row, row_w, dims, gamma, beta, running_mean, running_var, training, act,
                eps, momentum, out16=False, inv=None):
        ctx.inv = inv
        ps = _mat(ps)
        n, e, ne, count = dims
        co = ps.size(1) // 2
        dev = ps.device
        has_bn = gamma is not None
        ctx.has_bn, ctx.act, ctx.training, ctx.dims = has_bn, ACT[act], bool(training), dims
        idx = (ptr(grp_ptr), ptr(row_src), ptr(rep_row))
        par = None
        if has_bn:
            par = torch.empty((4, co), dtype=torch.float32, device=dev)
            if training:
                nparts = lib().ccn_cg_edge_stats_rows(n, co)
                partial = torch.empty((nparts + 1) * 2 * co, dtype=torch.float64, device=dev)
                call("cg_edge_stats", ptr(ps), _ld(ps), *idx, ptr(row_w), n, e, ne, co, ptr(partial))
                call("bn_finalize_n", ptr(partial), nparts, int(count), co, ptr(gamma), ptr(beta), float(eps),
                     float(momentum), ptr(running_mean), ptr(running_var), ptr(par[0]), ptr(par[1]), ptr(par[2]),
                     ptr(par[3]))
            else:
                call("bn_eval_params", ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var), float(eps), co,
                     ptr(par[0]), ptr(par[1]), ptr(par[2]), ptr(par[3]))
        ctx.save_for_backward(ps, grp_ptr, row_src, rep_row, row_w, par if has_bn else ps.new_empty(0))
        if out16:       # 16-bit storage modes: the plain last Linear of the MLP reads 16-bit rows (edge_out16)
            fdt = _fwd16()
            z = _rows16(e + ne + 1, co, dev, fdt)
            call("cg_edge_apply_h", ptr(ps), _ld(ps), *idx, n, e, ne, co, ptr(par[0]) if has_bn else None,
                 ptr(par[1]) if has_bn else None, ctx.act, LEAKY_SLOPE, ptr(z), _ld(z), 1 if fdt == torch.float16 else 0)
            return z.view(torch.bfloat16) if fdt == torch.float16 else z
        z = _rows(e + ne + 1, co, dev)
        call("cg_edge_apply", ptr(ps), _ld(ps), *idx, n, e, ne, co, ptr(par[0]) if has_bn else None,
             ptr(par[1]) if has_bn else None, ctx.act, LEAKY_SLOPE, ptr(z), _ld(z))
        _trace_act(z, ctx.act)
        return z

    @staticmethod
    def backward(ctx, g):
        ps, grp_ptr, row_src, rep_row, row_w, par = ctx.saved_tensors
        h = "_h" if _is_rows16(g) else ""          # the gradient of a 16-bit activation arrives as bf16 rows
        g = g if h else _mat(g.float() if g.dtype != torch.float32 else g)
        n, e, ne, count = ctx.dims
        co = ps.size(1) // 2
        dev = g.device
        idx = (ptr(grp_ptr), ptr(row_src), ptr(rep_row))
        sums = dgamma = dbeta = None
        pp = [None] * 4
        if ctx.inv is not None:
            # round 5: no atomics, one pass over dZ per index order (ccn_cg_edge_bwd_sums / _gather / _finish)
            inv_ptr, inv_row, row_dst = ctx.inv
            if ctx.has_bn:
                tab = par
            else:       # no BatchNorm: identity table
                tab = torch.zeros((4, co), dtype=torch.float32, device=dev)
                tab[0].fill_(1.0)
            pp = [ptr(tab[0]), ptr(tab[1]), ptr(tab[2]), ptr(tab[3])]
            dz16 = 1 if h else 0
            nparts = lib().ccn_cg_edge_stats_rows(n, co)
            partial = torch.empty((nparts + 1) * 2 * co, dtype=torch.float64, device=dev)
            pt, pq = _rows(n, 2 * co, dev), _rows(n, 2 * co, dev)
            call("cg_edge_bwd_sums", ptr(ps), _ld(ps), *idx, ptr(row_w), n, e, ne, co, ptr(g), dz16, _ld(g), *pp, ctx.act,
                 LEAKY_SLOPE, ptr(partial), ptr(pt), _ld(pt))
            call("cg_edge_bwd_gather", ptr(ps), _ld(ps), ptr(inv_ptr), ptr(inv_row), ptr(row_dst), n, co, ptr(g), dz16, _ld(g),
                 *pp, ctx.act, LEAKY_SLOPE, ptr(pq), _ld(pq), work_rows=e)
            sums = partial[nparts * 2 * co:]
            call("reduce_partials", ptr(partial), nparts, 2 * co, ptr(sums))
            if ctx.has_bn:
                dgb = sums[:2 * co].float()
                dbeta, dgamma = dgb[:co], dgb[co:]
            dps = _rows(ps.size(0), 2 * co, dev)
            if ps.size(0) > n:
                dps[n:].zero_()
            call("cg_edge_bwd_finish", ptr(pt), _ld(pt), ptr(pq), _ld(pq), ptr(grp_ptr), ptr(rep_row), ptr(row_w), ptr(inv_ptr),
                 n, e, co, pp[0], ptr(sums), float(count), 1 if (ctx.training and ctx.has_bn) else 0, ptr(dps), _ld(dps))
            return (dps,) + (None,) * 5 + (dgamma, dbeta) + (None,) * 8
        if ctx.has_bn:
            pp = [ptr(par[0]), ptr(par[1]), ptr(par[2]), ptr(par[3])]
            nparts = lib().ccn_cg_edge_stats_rows(n, co)
            partial = torch.empty((nparts + 1) * 2 * co, dtype=torch.float64, device=dev)
            call("cg_edge_bwd_stats" + h, ptr(ps), _ld(ps), *idx, ptr(row_w), n, e, ne, co, ptr(g), _ld(g), *pp, ctx.act,
                 LEAKY_SLOPE, ptr(partial))
            sums = partial[nparts * 2 * co:]
            call("reduce_partials", ptr(partial), nparts, 2 * co, ptr(sums))
            dgb = sums[:2 * co].float()                   # one conversion, two views
            dbeta, dgamma = dgb[:co], dgb[co:]
        dps = _rows(ps.size(0), 2 * co, dev)        # (the entry point zeroes the table it accumulates into)
        if ps.size(0) > n:
            dps[n:].zero_()
        call("cg_edge_bwd" + h, ptr(ps), _ld(ps), *idx, ptr(row_w), n, e, co, ptr(g), _ld(g), *pp, ctx.act, LEAKY_SLOPE,
             ptr(sums) if sums is not None else None, float(count), 1 if (ctx.training and ctx.has_bn) else 0, ptr(dps),
             _ld(dps))
        return (dps,) + (None,) * 5 + (dgamma, dbeta) + (None,) * 8


def cg_edge_layer(ps, comp, bn, training, act, out16=False):
    dims = (comp.n, comp.e, comp.ne, comp.count)
    inv = getattr(comp, "inv", None)
    if bn is None:
        return CGEdgeLayer.apply(ps, *comp.tensors(), dims, None, None, None, None, False, None, 0.0, 0.0, False, inv)
    if training and bn.track_running_stats:
        bn.num_batches_tracked += 1
    use_batch_stats = training or not bn.track_running_stats
    return _mark16(CGEdgeLayer.apply(ps, *comp.tensors(), dims, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                     use_batch_stats, act, bn.eps, bn.momentum if bn.momentum is not None else 0.1,
                                     bool(out16), inv), out16)


class LinearBNActTail(torch.autograd.Function):
    """``LinearBNAct`` (no bias) over rows whose tail [tail:] carries weights: BatchNorm statistics, its backward
    reductions and the weight gradient count row r of the tail ``w[r - tail]`` times (total ``count`` rows)."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, running_mean, running_var, training, act, eps, momentum, tail, w, count,
                grad_on=True):
        x = _mat(x)
        m, k = x.shape
        n = weight.size(0)
        dev = x.device
        wt = _aligned_weight(weight.detach())
        ctx.act, ctx.training, ctx.tail, ctx.count = ACT[act], bool(training), int(tail), float(count)
        ctx.main_grad_of = weight if (grad_on and ctx.needs_input_grad[1] and _main_grad(weight, n, k) is not None) else None
        _main_grad_note(ctx.main_grad_of)
        gemm_nt = ctx.gemm_nt = _nt_name()
        if gemm_nt != "gemm_nt":
            x = _aligned_rows(x)
        y = _rows(m, n, dev)
        t = m - tail
        xt, yt = x[tail:], y[tail:]
        par = torch.empty((4, n), dtype=torch.float32, device=dev)
        if training:
            nparts = lib().ccn_stats_rows(tail)
            stats = torch.empty((nparts + 2) * 2 * n, dtype=torch.float64, device=dev)
            _gemm_nt(gemm_nt, x, wt, None, y, tail, n, k, stats)
            _gemm_nt(gemm_nt, x, wt, None, y, t, n, k, None, xp=ptr(xt), yp=ptr(yt))
            acc = torch.empty((lib().ccn_stats_rows(t) + 1) * 2 * n, dtype=torch.float64, device=dev)
            call("colstats_weighted", ptr(yt), _ld(y), ptr(w), t, n, ptr(acc))
            stats[nparts * 2 * n:(nparts + 1) * 2 * n].copy_(acc[:2 * n])          # one more partial row
            call("bn_finalize_n", ptr(stats), nparts + 1, int(count), n, ptr(gamma), ptr(beta), float(eps),
                 float(momentum), ptr(running_mean), ptr(running_var), ptr(par[0]), ptr(par[1]), ptr(par[2]), ptr(par[3]))
        else:
            _gemm_nt(gemm_nt, x, wt, None, y, m, n, k, None)
            call("bn_eval_params", ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var), float(eps), n,
                 ptr(par[0]), ptr(par[1]), ptr(par[2]), ptr(par[3]))
        z = _rows(m, n, dev)
        call("bn_act_fwd", ptr(y), _ld(y), m, n, ptr(par[0]), ptr(par[1]), ctx.act, LEAKY_SLOPE, ptr(z), _ld(z))
        ctx.save_for_backward(x, wt, y, par, w)
        _trace_act(z, ctx.act)
        return z

    @staticmethod
    def backward(ctx, g):
        x, wt, y, par, w = ctx.saved_tensors
        g = _mat(g)
        dev = g.device
        m, n = y.shape
        k = x.size(1)
        tail = ctx.tail
        t = m - tail
        pp = (ptr(par[0]), ptr(par[1]), ptr(par[2]), ptr(par[3]))
        sums = _stats_buffer(tail, n, dev)
        call("bn_act_bwd_reduce", ptr(g), _ld(g), ptr(y), _ld(y), tail, n, *pp, ctx.act, LEAKY_SLOPE, ptr(sums))
        sums_t = _stats_buffer(t, n, dev)
        gt, yt = g[tail:], y[tail:]
        call("bn_act_bwd_reduce_weighted", ptr(gt), _ld(g), ptr(yt), _ld(y), ptr(w), t, n, *pp, ctx.act, LEAKY_SLOPE,
             ptr(sums_t))
        sums[:2 * n].add_(sums_t[:2 * n])
        dy = _rows(m, n, dev)
        dgb = torch.empty((2, n), dtype=torch.float32, device=dev)
        call("bn_act_bwd_apply_count", ptr(g), _ld(g), ptr(y), _ld(y), m, n, *pp, ctx.act, LEAKY_SLOPE, ptr(sums),
             ctx.count, 1 if ctx.training else 0, ptr(dy), _ld(dy), ptr(dgb[0]), ptr(dgb[1]))
        dx = None
        if ctx.needs_input_grad[0]:
            dx = _rows(m, k, dev)
            wtt = _transposed(wt, n, k)
            _gemm_nt(_GEMM_BWD[ctx.gemm_nt], dy, wtt, None, dx, m, k, n, None)
        into = _main_grad(ctx.main_grad_of, n, k)
        if into is None and ctx.main_grad_of is not None:
            _main_grad_cancel(ctx.main_grad_of)
        dw = into if into is not None else _rows(n, k, dev, zero=True)
        dyt = _rows(t, n, dev)
        torch.mul(dy[tail:], w[:, None], out=dyt)
        xt = x[tail:]
        with _WgradScope(into, dy, x, dyt):
            _wgrad(ctx.gemm_nt, dy, x, dw, tail, n, k)
            call("gemm_tn", ptr(dyt), _ld(dyt), ptr(xt), _ld(x), ptr(dw), _ld(dw), t, n, k)   # few rows: fp32
        if into is not None:
            dw = _main_grad_done(ctx.main_grad_of)
        return dx, dw, dgb[0], dgb[1], None, None, None, None, None, None, None, None, None, None


def linear_bn_act_tail(x, weight, bn, training, act, tail, w, count):
    if training and bn.track_running_stats:
        bn.num_batches_tracked += 1
    use_batch_stats = training or not bn.track_running_stats
    return LinearBNActTail.apply(x, weight, bn.weight, bn.bias, bn.running_mean, bn.running_var, use_batch_stats, act,
                                 bn.eps, bn.momentum if bn.momentum is not None else 0.1, tail, w, count,
                                 torch.is_grad_enabled())


class CGMax(torch.autograd.Function):
    """``SGMax`` on compact rows (ref dgcnn.py:181,187-189,206)."""

    @staticmethod
    def forward(ctx, f, grp_ptr, rep_row, n, row_src=None):
        f = _mat(f)
        c = f.size(1)
        out = _rows(n, c, f.device)
        arg = torch.empty((n, c), dtype=torch.int32, device=f.device)
        call("cg_max_fwd", ptr(f), _ld(f), ptr(grp_ptr), ptr(rep_row), n, c, ptr(out), _ld(out), ptr(arg), work_rows=f.size(0))
        ctx.save_for_backward(arg, grp_ptr, rep_row)
        ctx.shape = (n, f.size(0), c)
        if MAX_TRACE is not None and row_src is not None:
            _trace_max(arg, grp_ptr[:-1], row_src)
        return out

    @staticmethod
    def backward(ctx, g):
        arg, grp_ptr, rep_row = ctx.saved_tensors
        g = _mat(g)
        n, rows, c = ctx.shape
        df = _rows(rows, c, g.device)
        call("cg_max_bwd", ptr(g), _ld(g), ptr(arg), ptr(grp_ptr), ptr(rep_row), n, rows, c, ptr(df), _ld(df))
        return df, None, None, None, None


class ShiftAddBNAct(torch.autograd.Function):
    """Second half of a curve-conv layer computed as "product first, shift-add second" (see ccn_shift_add_fwd):
    y = shift_add(P) + b, BatchNorm over the rows (batch statistics in training mode), activation."""

    @staticmethod
    def forward(ctx, p, bias, gamma, beta, running_mean, running_var, training, act, eps, momentum, taps):
        p = _mat(p)
        m = p.size(0)
        n = p.size(1) // taps
        dev = p.device
        ctx.act, ctx.training, ctx.taps, ctx.has_bias, ctx.ldp = ACT[act], bool(training), taps, bias is not None, p.size(1)
        y = _rows(m, n, dev)
        call("shift_add_fwd", ptr(p), _ld(p), ptr(bias), m, n, taps, ptr(y), _ld(y))
        par = torch.empty((4, n), dtype=torch.float32, device=dev)
        if training:
            if m < 2:
                raise ValueError("Expected more than 1 value per channel when training")
            acc = _stats_buffer(m, n, dev)
            call("colstats_weighted", ptr(y), _ld(y), None, m, n, ptr(acc))
            # acc[0:2n] holds the totals: finalise from that single row
            call("bn_finalize_n", ptr(acc), 1, m, n, ptr(gamma), ptr(beta), float(eps), float(momentum), ptr(running_mean),
                 ptr(running_var), ptr(par[0]), ptr(par[1]), ptr(par[2]), ptr(par[3]))
        else:
            call("bn_eval_params", ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var), float(eps), n,
                 ptr(par[0]), ptr(par[1]), ptr(par[2]), ptr(par[3]))
        z = _rows(m, n, dev)
        call("bn_act_fwd", ptr(y), _ld(y), m, n, ptr(par[0]), ptr(par[1]), ctx.act, LEAKY_SLOPE, ptr(z), _ld(z))
        ctx.save_for_backward(y, par)
        _trace_act(z, ctx.act)
        return z

    @staticmethod
    def backward(ctx, g):
        y, par = ctx.saved_tensors
        g = _mat(g)
        dev = g.device
        m, n = y.shape
        pp = (ptr(par[0]), ptr(par[1]), ptr(par[2]), ptr(par[3]))
        sums = _stats_buffer(m, n, dev)
        call("bn_act_bwd_reduce", ptr(g), _ld(g), ptr(y), _ld(y), m, n, *pp, ctx.act, LEAKY_SLOPE, ptr(sums))
        dy = _rows(m, n, dev)
        dgb = torch.empty((2, n), dtype=torch.float32, device=dev)
        call("bn_act_bwd_apply", ptr(g), _ld(g), ptr(y), _ld(y), m, n, *pp, ctx.act, LEAKY_SLOPE, ptr(sums),
             1 if ctx.training else 0, ptr(dy), _ld(dy), ptr(dgb[0]), ptr(dgb[1]))
        dp = _rows(m, ctx.ldp, dev)
        call("shift_add_bwd", ptr(dy), _ld(dy), m, n, ctx.taps, ptr(dp), _ld(dp))
        db = None
        if ctx.has_bias:
            acc = _stats_buffer(m, n, dev)
            db = torch.empty(n, dtype=torch.float32, device=dev)
            call("colsum", ptr(dy), _ld(dy), m, n, ptr(acc), ptr(db))
        return dp, db, dgb[0], dgb[1], None, None, None, None, None, None, None


class ConvRowsBNAct(torch.autograd.Function):
    """One conv + BatchNorm + activation layer of the curve convolutions as an IMPLICIT GEMM over the row sequence
    (ccn_conv_rows_nt / _tn, include/ccn_hip.h): no shifted-row matrix is materialised -- row i of it is the contiguous
    span of ``taps`` consecutive sequence rows, which the GEMM loaders read in place from a buffer with taps // 2 zero
    halo rows at both ends (ref fast_conv1d.py:71-73 on the zero-separated sequence of :48-61; :136-143 for V1).

    ``excl`` (int64 row numbers, V1): the zero separator rows between curves.  They take part in the convolution as zeros
    but NOT in the BatchNorm (the reference normalises the N real rows only, quirk Q1): their share is subtracted from the
    batch statistics, their output and their gradients are zeroed."""

    @staticmethod
    def forward(ctx, x, gw, bias, gamma, beta, running_mean, running_var, training, act, eps, momentum, taps, excl):
        x = _mat(x)
        rows, cin = x.shape
        cout = gw.size(0)
        h = taps // 2
        dev = x.device
        if gw.size(1) != taps * cin:
            raise ValueError("conv: input has %d channels, weight expects %d" % (cin, gw.size(1) // taps))
        if _halo_of(x, h) is None:              # producer did not leave a halo: one copy (still no taps-wide matrix)
            xh = _rows_halo(rows, cin, h, dev)
            xh.copy_(x)
            x = xh
        buf, hh = _halo_of(x, h)
        ld = buf.stride(0)
        a_ptr = ctypes.c_void_p(x.data_ptr() - h * ld * 4)
        k = taps * ld
        wp = torch.zeros((cout, taps, ld), dtype=torch.float32, device=dev)
        wp[:, :, :cin] = gw.detach().view(cout, taps, cin)
        ctx.act, ctx.training, ctx.taps, ctx.has_bias, ctx.cin = ACT[act], bool(training), taps, bias is not None, cin
        n_excl = 0 if excl is None else excl.numel()
        ctx.count = float(rows - n_excl)
        y = _rows(rows, cout, dev)
        par = torch.empty((4, cout), dtype=torch.float32, device=dev)
        if training:
            if rows - n_excl < 2:
                raise ValueError("Expected more than 1 value per channel when training")
            nparts = lib().ccn_stats_rows(rows)
            stats = torch.zeros((nparts + 2) * 2 * cout, dtype=torch.float64, device=dev)
            call("conv_rows_nt", a_ptr, ld, ptr(wp), k, ptr(bias), ptr(y), _ld(y), rows, cout, k, ptr(stats))
            if n_excl:
                ye = y.index_select(0, excl)
                acc = torch.empty((lib().ccn_stats_rows(n_excl) + 1) * 2 * cout, dtype=torch.float64, device=dev)
                call("colstats_weighted", ptr(ye), _ld(ye), None, n_excl, cout, ptr(acc))
                stats[nparts * 2 * cout:(nparts + 1) * 2 * cout] = -acc[:2 * cout]     # one more (negative) partial row
            call("bn_finalize_n", ptr(stats), nparts + 1, int(ctx.count), cout, ptr(gamma), ptr(beta), float(eps),
                 float(momentum), ptr(running_mean), ptr(running_var), ptr(par[0]), ptr(par[1]), ptr(par[2]), ptr(par[3]))
        else:
            call("conv_rows_nt", a_ptr, ld, ptr(wp), k, ptr(bias), ptr(y), _ld(y), rows, cout, k, None)
            call("bn_eval_params", ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var), float(eps), cout,
                 ptr(par[0]), ptr(par[1]), ptr(par[2]), ptr(par[3]))
        z = _rows_halo(rows, cout, h, dev)       # the next layer convolves it in place
        call("bn_act_fwd", ptr(y), _ld(y), rows, cout, ptr(par[0]), ptr(par[1]), ctx.act, LEAKY_SLOPE, ptr(z), _ld(z))
        if n_excl:
            z.index_fill_(0, excl, 0.0)
        ctx.save_for_backward(x, wp, y, par, excl if n_excl else x.new_empty(0, dtype=torch.int64))
        _trace_act(z, ctx.act)
        return z

    @staticmethod
    def backward(ctx, g):
        x, wp, y, par, excl = ctx.saved_tensors
        g = _mat(g)
        dev = g.device
        rows, cout = y.shape
        taps, cin = ctx.taps, ctx.cin
        h = taps // 2
        ld = x.stride(0)
        k = taps * ld
        pp = (ptr(par[0]), ptr(par[1]), ptr(par[2]), ptr(par[3]))
        if excl.numel():
            g = g.index_fill(0, excl, 0.0)       # separator rows are constants
        sums = _stats_buffer(rows, cout, dev)
        call("bn_act_bwd_reduce", ptr(g), _ld(g), ptr(y), _ld(y), rows, cout, *pp, ctx.act, LEAKY_SLOPE, ptr(sums))
        dy = _rows_halo(rows, cout, h, dev)
        dgb = torch.empty((2, cout), dtype=torch.float32, device=dev)
        call("bn_act_bwd_apply_count", ptr(g), _ld(g), ptr(y), _ld(y), rows, cout, *pp, ctx.act, LEAKY_SLOPE, ptr(sums),
             ctx.count, 1 if ctx.training else 0, ptr(dy), _ld(dy), ptr(dgb[0]), ptr(dgb[1]))
        if excl.numel():
            dy.index_fill_(0, excl, 0.0)
        ldo = dy.stride(0)
        dx = None
        if ctx.needs_input_grad[0]:
            # dX[j] = sum_t dY[j + h - t] W_t: the same implicit GEMM over dY with the taps reversed
            wf = torch.zeros((cin, taps, ldo), dtype=torch.float32, device=dev)
            wf[:, :, :cout] = wp[:, :, :cin].flip(1).permute(2, 1, 0)
            dx = _rows(rows, cin, dev)
            call("conv_rows_nt", ctypes.c_void_p(dy.data_ptr() - h * ldo * 4), ldo, ptr(wf), taps * ldo, None, ptr(dx),
                 _ld(dx), rows, cin, taps * ldo, None)
        dgw = None
        if ctx.needs_input_grad[1]:
            dw = torch.zeros((cout, k), dtype=torch.float32, device=dev)
            nb = lib().ccn_gemm_tn_workspace_bytes(rows, cout, k)
            ws = _tn_scratch(nb, dev)
            call("conv_rows_tn", ptr(dy), ldo, ctypes.c_void_p(x.data_ptr() - h * ld * 4), ld, ptr(dw), k, rows, cout, k,
                 ptr(ws), nb)
            dgw = dw.view(cout, taps, ld)[:, :, :cin].reshape(cout, taps * cin)
        db = None
        if ctx.has_bias:
            acc = _stats_buffer(rows, cout, dev)
            db = torch.empty(cout, dtype=torch.float32, device=dev)
            call("colsum", ptr(dy), _ld(dy), rows, cout, ptr(acc), ptr(db))
        return dx, dgw, db, dgb[0], dgb[1], None, None, None, None, None, None, None, None


class ConvRowsBNActH(torch.autograd.Function):
    """ConvRowsBNAct for the 16-bit storage modes (r4): the implicit GEMM runs on a 16-BIT copy of the row sequence
    (ccn_conv_rows_nt_h / _tn_h) instead of on a materialised 16-bit shifted-row matrix -- one cast of the sequence (6 B per
    element) where ccn_im2col_fwd_h moved 4 + 2 * taps, and no col2im pass in backward: the data gradient is the same implicit
    product over the bf16 dY sequence with the taps reversed.  Operands rounded exactly as the shifted-row form rounds them
    (forward: bf16 / fp16 of x and W; backward: bf16 of dY, of W and of the forward operand), fp32 accumulation; the
    activation between layers stays an fp32 sequence (DiffConcat runs on it).  Same ``excl`` semantics as ConvRowsBNAct."""

    @staticmethod
    def forward(ctx, x, gw, bias, gamma, beta, running_mean, running_var, training, act, eps, momentum, taps, excl):
        x = _mat(x)
        rows, cin = x.shape
        cout = gw.size(0)
        h = taps // 2
        dev = x.device
        if gw.size(1) != taps * cin:
            raise ValueError("conv: input has %d channels, weight expects %d" % (cin, gw.size(1) // taps))
        if _halo_of(x, h) is None:
            xh = _rows_halo(rows, cin, h, dev)
            xh.copy_(x)
            x = xh
        buf, hh = _halo_of(x, h)
        fdt = _fwd16()
        f16 = 1 if fdt == torch.float16 else 0
        ld16 = (cin + 7) // 8 * 8
        # the sequence INCLUDING its halo rows as 16-bit rows (padding columns zeroed by the cast)
        x16 = torch.empty((rows + 2 * h, ld16), dtype=fdt, device=dev)
        first = buf[hh - h:hh + rows + h]
        call("cast_rows_h", ptr(first), first.stride(0), rows + 2 * h, cin, ptr(x16), ld16, f16)
        k = taps * ld16
        wp = torch.zeros((cout, taps, ld16), dtype=torch.float32, device=dev)
        wp[:, :, :cin] = gw.detach().view(cout, taps, cin)
        wp16 = _cast16(wp.view(cout, k), fdt)
        ctx.act, ctx.training, ctx.taps, ctx.has_bias, ctx.cin = ACT[act], bool(training), taps, bias is not None, cin
        ctx.f16 = f16
        n_excl = 0 if excl is None else excl.numel()
        ctx.count = float(rows - n_excl)
        y = _rows(rows, cout, dev)
        par = torch.empty((4, cout), dtype=torch.float32, device=dev)
        a_ptr = ptr(x16)                       # (row 0 of x16 is the first halo row: the span of row i starts at row i)
        if training:
            if rows - n_excl < 2:
                raise ValueError("Expected more than 1 value per channel when training")
            nparts = lib().ccn_stats_rows(rows)
            stats = torch.zeros((nparts + 2) * 2 * cout, dtype=torch.float64, device=dev)
            call("conv_rows_nt_h", a_ptr, ld16, ptr(wp16), _ld(wp16), ptr(bias), ptr(y), _ld(y), rows, cout, k, ptr(stats), f16, 0)
            if n_excl:
                ye = y.index_select(0, excl)
                acc = torch.empty((lib().ccn_stats_rows(n_excl) + 1) * 2 * cout, dtype=torch.float64, device=dev)
                call("colstats_weighted", ptr(ye), _ld(ye), None, n_excl, cout, ptr(acc))
                stats[nparts * 2 * cout:(nparts + 1) * 2 * cout] = -acc[:2 * cout]
            call("bn_finalize_n", ptr(stats), nparts + 1, int(ctx.count), cout, ptr(gamma), ptr(beta), float(eps),
                 float(momentum), ptr(running_mean), ptr(running_var), ptr(par[0]), ptr(par[1]), ptr(par[2]), ptr(par[3]))
        else:
            call("conv_rows_nt_h", a_ptr, ld16, ptr(wp16), _ld(wp16), ptr(bias), ptr(y), _ld(y), rows, cout, k, None, f16, 0)
            call("bn_eval_params", ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var), float(eps), cout,
                 ptr(par[0]), ptr(par[1]), ptr(par[2]), ptr(par[3]))
        z = _rows_halo(rows, cout, h, dev)
        call("bn_act_fwd", ptr(y), _ld(y), rows, cout, ptr(par[0]), ptr(par[1]), ctx.act, LEAKY_SLOPE, ptr(z), _ld(z))
        if n_excl:
            z.index_fill_(0, excl, 0.0)
        ctx.save_for_backward(x16, wp, y, par, excl if n_excl else x.new_empty(0, dtype=torch.int64))
        _trace_act(z, ctx.act)
        return z

    @staticmethod
    def backward(ctx, g):
        x16, wp, y, par, excl = ctx.saved_tensors
        g = _mat(g)
        dev = g.device
        rows, cout = y.shape
        taps, cin = ctx.taps, ctx.cin
        h = taps // 2
        ld16 = x16.stride(0)
        k = taps * ld16
        pp = (ptr(par[0]), ptr(par[1]), ptr(par[2]), ptr(par[3]))
        if excl.numel():
            g = g.index_fill(0, excl, 0.0)
        sums = _stats_buffer(rows, cout, dev)
        call("bn_act_bwd_reduce", ptr(g), _ld(g), ptr(y), _ld(y), rows, cout, *pp, ctx.act, LEAKY_SLOPE, ptr(sums))
        ldo = (cout + 7) // 8 * 8
        dy16 = torch.zeros((rows + 2 * h, ldo), dtype=torch.bfloat16, device=dev)     # bf16 dY sequence, zero halo and padding
        dyv = dy16[h:h + rows]
        dgb = torch.empty((2, cout), dtype=torch.float32, device=dev)
        call("bn_act_bwd_apply_h", ptr(g), 0, _ld(g), ptr(y), _ld(y), rows, cout, *pp, ctx.act, LEAKY_SLOPE, ptr(sums),
             ctx.count, 1 if ctx.training else 0, 0, ptr(dyv), ldo, ptr(dgb[0]), ptr(dgb[1]), 0)
        if excl.numel():
            dyv.index_fill_(0, excl, 0.0)
        dx = None
        if ctx.needs_input_grad[0]:
            wf = torch.zeros((cin, taps, ldo), dtype=torch.float32, device=dev)
            wf[:, :, :cout] = wp[:, :, :cin].flip(1).permute(2, 1, 0)
            wf16 = _cast16(wf.view(cin, taps * ldo))
            dx = _rows(rows, cin, dev)
            call("conv_rows_nt_h", ptr(dy16), ldo, ptr(wf16), _ld(wf16), None, ptr(dx), _ld(dx), rows, cin, taps * ldo, None, 0, 0)
        dgw = None
        if ctx.needs_input_grad[1]:
            dw = torch.zeros((cout, k), dtype=torch.float32, device=dev)
            nb = lib().ccn_gemm_tn_h_workspace_bytes(rows, cout, k)
            ws = _tn_scratch(nb, dev)
            call("conv_rows_tn_h", ptr(dyv), ldo, ptr(x16), ld16, ctx.f16, ptr(dw), k, rows, cout, k, ptr(ws), nb)
            dgw = dw.view(cout, taps, ld16)[:, :, :cin].reshape(cout, taps * cin)
        db = None
        if ctx.has_bias:
            # the bias sits in front of the BatchNorm: with batch statistics its gradient is identically zero (the column sums
            # of the bf16 dY rows would be rounding noise), with running statistics it is scale * sum(g act') from the first pass
            db = torch.zeros(cout, dtype=torch.float32, device=dev) if ctx.training else par[0] * sums[:cout].float()
        return dx, dgw, db, dgb[0], dgb[1], None, None, None, None, None, None, None, None


def conv_rows_implicit(x, gemm_weight, bias, bn, training, act, taps, excl=None):
    """conv + BatchNorm + activation over a row sequence without the shifted-row matrix (ConvRowsBNAct; ConvRowsBNActH in
    the 16-bit storage modes)."""
    if training and bn.track_running_stats:
        bn.num_batches_tracked += 1
    use_batch_stats = training or not bn.track_running_stats
    fn = ConvRowsBNActH if conv_implicit_16bit() else ConvRowsBNAct
    return fn.apply(x, gemm_weight, bias, bn.weight, bn.bias, bn.running_mean, bn.running_var, use_batch_stats,
                    act, bn.eps, bn.momentum if bn.momentum is not None else 0.1, taps, excl)


CONV_IMPLICIT_H = os.environ.get("CCN_CONV_IMPLICIT_H", "1") != "0"     # (A/B and tests: 0 = the 16-bit shifted-row matrix)


def conv_implicit_16bit():
    """The 16-bit storage modes run the curve convolutions as implicit GEMMs on a 16-bit sequence (ConvRowsBNActH)."""
    return bool(CONV_IMPLICIT_H and EDGE_OUT16 and STORE16 and _MLP_DTYPE in ("bf16", "fp16") and ACT_TRACE is None)


def conv_rows_bn_act(x, gemm_weight, bias, bn, training, act, taps):
    """One conv + BatchNorm + activation layer on an unsegmented row sequence (V2 layout) for C_in >= 2*C_out:
    P = X W_all^T over the rows, then the shift-add; ``gemm_weight`` is the (C_out, taps*C_in) shifted-row matrix."""
    co, cin = gemm_weight.size(0), gemm_weight.size(1) // taps
    w_all = gemm_weight.view(co, taps, cin).permute(1, 0, 2).reshape(taps * co, cin)
    p = linear_bn_act(x, w_all, None, None, False, None)
    if training and bn.track_running_stats:
        bn.num_batches_tracked += 1
    use_batch_stats = training or not bn.track_running_stats
    return ShiftAddBNAct.apply(p, bias, bn.weight, bn.bias, bn.running_mean, bn.running_var, use_batch_stats, act, bn.eps,
                               bn.momentum if bn.momentum is not None else 0.1, taps)
