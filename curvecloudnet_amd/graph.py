"""hipGraph capture of the feature pass (BASELINE configs[4]: "hipGraph-captured fwd").

A forward pass splits into the position-only part (sampling, neighbour search, index tables: data-dependent sizes, host
read-backs) and the feature part (every GEMM / BatchNorm / gather kernel: sizes fixed once the tables exist).
``ModelBase.prepare(data)`` computes the first ahead of time; ``CapturedForward`` records the second -- a few hundred to
a few thousand ``ccn_*`` launches -- into ONE graph and replays it with a single host call.  Every entry point of
``libccn_hip.so`` takes its stream as an argument, allocates nothing and never synchronises, so the whole feature pass is
capturable; PyTorch supplies the capture plumbing (``torch.cuda.CUDAGraph`` = hipGraph on ROCm, with the caching
allocator's private pool for the intermediates).

Use: inference over a batch whose geometry is prepared (the same clouds evaluated repeatedly -- test-time augmentation of
features, ensembles of weights -- or a fixed-topology stream).  The logits of a replay are bit-identical to the eager
pass (tests/test_gpu_graph.py).
"""
import contextlib

import torch


@contextlib.contextmanager
def _generator_state(state):
    """Run the enclosed pass from the CPU generator state ``state`` and put the caller's state back afterwards: the samplers' draws
    of a captured forward are those of construction time, but the caller's own stream of random numbers (augmentation, shuffling,
    dropout seeding) must not be rewound by it."""
    saved = torch.get_rng_state()
    torch.set_rng_state(state)
    try:
        yield
    finally:
        torch.set_rng_state(saved)


class CapturedForward:
    def __init__(self, model, data, plan=None, warmup=1, **forward_kwargs):
        if plan is None:
            plan = model.prepare(data)
        if plan is None:
            raise RuntimeError("graph capture needs the geometry side stream (CCN_GEOMETRY_STREAM != 0) and a model without "
                               "feature-space searches: ModelBase.prepare() returned no plan")
        ctx, tables, owner = plan
        if owner is not data:
            raise ValueError("the plan was prepared for another batch")
        # every index table must be complete before the capture starts: the graph carries no dependency on the geometry
        # stream (the events of the plan are dropped)
        ctx.side.synchronize()
        self.plan = (ctx, [None if t is None else (t[0], None) for t in tables], data)
        self.model, self.data, self.kwargs = model, data, forward_kwargs
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.no_grad():
            with torch.cuda.stream(side):                 # warm-up off the default stream, as capture requires
                for _ in range(warmup):
                    model(data, plan=self.plan, **forward_kwargs)
            cur.wait_stream(side)
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            from . import ops
            ops.NT_CAPTURE = self._scratch = {}            # (one tail-split scratch per capture, alive as long as the graph)
            try:
                with torch.cuda.graph(self.graph):
                    self.out = model(data, plan=self.plan, **forward_kwargs)
            finally:
                ops.NT_CAPTURE = None
        self.launches = None

    def eager(self):
        """The same pass launched kernel by kernel (for A/B timing and the bit-identity test)."""
        with torch.no_grad():
            return self.model(self.data, plan=self.plan, **self.kwargs)

    def replay(self):
        """One graph launch; returns the (static) logits tensor of the captured pass."""
        self.graph.replay()
        return self.out


class CapturedWholeForward:
    """The WHOLE inference forward -- sampling, neighbour searches, index tables AND the feature pass -- in one hipGraph
    (BASELINE configs[4] "hipGraph-captured fwd"; SURVEY.md section 7.8 / 8(b) "hipGraph-capturable").

    The reference turns every data-dependent size into a host integer with a device -> host synchronisation
    (``torch.where`` in ``batch2ptr``, point_ops.py:50; boolean flattening :101-107; ``fps_ops.py:31-33``), and so does this
    package by default: a graph cannot contain those.  Here the counts stay on the device (``ops.CountBounds``):

    1. a CALIBRATION forward over the batch logs every count in program order (``ops.CountRecorder``);
    2. each becomes a capacity (count x ``headroom``, rounded up); buffers are allocated and kernels launched for the
       capacities, the true counts live in the CSR offsets / group pointers / per-cloud lengths the kernels read anyway;
    3. the slack has to be SOMETHING: one phantom point is appended to the batch as a cloud of its own (cloud B, its own
       curve); index lists are padded with it, so every entry past a true count is one more point of that cloud -- the
       per-cloud searches, per-curve operations and per-row layers of the real clouds never see it, and in inference
       mode (BatchNorm from running statistics) nothing reduces over rows;
    4. ``overflow`` (one device flag) is raised when a true count exceeds its capacity or an ordering check fails:
       ``replay()`` reads it back -- the ONE read-back of a forward -- and raises ``CapacityExceeded``; the caller then runs
       the eager path (or re-captures with the new batch as calibration).

    A replay is valid for the captured batch (new FEATURES may be written into it: ``load_features``) and for any batch of
    the same shape that ``load()`` has checked against the capacities; the random draws of the samplers (CurveFPS phase ...)
    are those of construction time."""

    class CapacityExceeded(RuntimeError):
        pass

    def __init__(self, model, data, headroom=1.0625, point_capacity=None, **forward_kwargs):
        """``point_capacity`` (round 6, VERDICT r5 missing #2): rows of the captured point tensors.  Default: the batch's own point
        count + 1 -- only batches of exactly that size can be loaded later.  Larger: the batch is padded with phantom points (cloud
        B, isolated on a far lattice) up to the capacity, and ``load()`` admits ANY batch of at most ``point_capacity - 1`` points
        and the same number of clouds -- a stream of different clouds through one graph.  Size it for the largest batch expected
        (a few per cent above the typical one: every phantom point costs what an isolated real point costs)."""
        from . import ops
        if model.training:
            raise RuntimeError("CapturedWholeForward captures the inference forward: call model.eval() first")
        self.model, self.kwargs = model, forward_kwargs
        self.n = data.pos.size(0)
        self.capacity = int(point_capacity) if point_capacity is not None else self.n + 1
        if self.capacity < self.n + 1:
            raise ValueError("point_capacity %d cannot hold the %d points of the batch + one phantom point" % (self.capacity, self.n))
        dev = data.pos.device
        self.data = self._with_phantom(data, self.capacity - self.n)
        # the samplers' random draws (CurveFPS phase, FPS starts ...) come from torch's CPU generator and end up as kernel
        # ARGUMENTS, i.e. inside the graph: every pass of this object starts from the generator state of construction time
        self._rng = torch.get_rng_state()
        # 1. the ordinary forward (reference value for the tests); its counts and its samplers' random draws are logged
        with ops.counts_scope(ops.CountRecorder()) as rec, torch.no_grad(), _generator_state(self._rng):
            self.reference = model(self.data, **forward_kwargs)[: self.n].clone()
        self.draws = rec.draws
        # 2. calibration: one eager pass in which every count is read back, turned into a capacity and used as such at
        # once (ops.CountBounds); afterwards the same pass without any read-back, off the default stream (allocator warm-up)
        self.bounds = ops.CountBounds(None, dev, headroom, draws=self.draws)
        with ops.counts_scope(self.bounds), torch.no_grad(), _generator_state(self._rng):
            model(self.data, **forward_kwargs)
        self.counts, self.caps = list(self.bounds.counts), list(self.bounds.caps)
        cur = torch.cuda.current_stream(dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            self.bounded_eager()
        cur.wait_stream(side)
        torch.cuda.synchronize(dev)
        # 3. capture
        self.graph = torch.cuda.CUDAGraph()
        ops.NT_CAPTURE = self._scratch = {}                # (one tail-split scratch per capture, alive as long as the graph)
        try:
            with ops.counts_scope(self.bounds), _generator_state(self._rng):
                with torch.no_grad(), torch.cuda.graph(self.graph):
                    self.bounds.rewind()
                    self.out = model(self.data, **forward_kwargs)
        finally:
            ops.NT_CAPTURE = None

    @staticmethod
    def _phantom_positions(anchor, count, device, dtype):
        """``count`` isolated positions: a lattice of spacing 16 (no radius of a shipped section exceeds 0.8) starting 16 away from
        ``anchor`` -- the cheapest neighbourhoods there are, and never inside a real cloud's grid."""
        j = torch.arange(1, count + 1, device=device)
        lattice = torch.stack([(j % 64), (j // 64) % 64, j // 4096], dim=1).to(dtype) * 16.0
        return anchor.to(dtype).view(1, 3) + lattice

    @staticmethod
    def _with_phantom(data, pad=1):
        """The batch + ``pad`` phantom points: cloud id B, ONE curve of that cloud, zero features, isolated positions (the first one
        at the position of the last real point, as in round 5: a single phantom keeps its old place)."""
        from types import SimpleNamespace
        out = SimpleNamespace(**vars(data))
        b = int(getattr(data, "num_clouds", None) or getattr(data, "num_graphs", None) or (int(data.batch[-1]) + 1))
        dev = data.pos.device
        ppos = data.pos[-1:] if pad == 1 else torch.cat(
            [data.pos[-1:], CapturedWholeForward._phantom_positions(data.pos[-1], pad - 1, dev, data.pos.dtype)], 0)
        out.pos = torch.cat([data.pos, ppos], 0).contiguous()
        out.batch = torch.cat([data.batch, torch.full((pad,), b, dtype=data.batch.dtype, device=dev)])
        out.curve_idxs = torch.cat([data.curve_idxs, torch.zeros(pad, dtype=data.curve_idxs.dtype, device=dev)])
        if getattr(data, "x", None) is not None:
            out.x = torch.cat([data.x, torch.zeros((pad,) + tuple(data.x.shape[1:]), dtype=data.x.dtype, device=dev)], 0).contiguous()
        if hasattr(data, "labels"):
            out.labels = torch.cat([data.labels, data.labels[-1:]])
        out.num_clouds = b + 1
        if hasattr(out, "num_graphs"):
            out.num_graphs = b + 1
        return out

    def load_features(self, x):
        """New point features for the captured batch (same positions: the counts cannot change, no check needed)."""
        self.data.x[: self.n].copy_(x)

    def load(self, data, verify=True):
        """Write another batch into the captured input tensors: at most ``point_capacity - 1`` points (exactly the captured count when
        no capacity was given), the same number of clouds; the rows behind it become phantom points.  New positions mean new counts.
        ``verify=True`` runs ONE eager pass that reads every count back and raises ``CapacityExceeded`` before a count that does not
        fit is used (ops.CountBounds, verifying) -- after a successful check every replay over this batch is safe.
        ``verify=False`` skips that pass (a stream of clouds: no extra pass, no read-back).  A count past its capacity then raises
        the device flag and ``replay()`` raises ``CapacityExceeded`` -- afterwards; the replay itself stays inside its buffers: fill
        kernels stop at the capacities, offsets / group pointers / per-cloud lengths are clamped to them, tables the bounded
        launches do not cover are initialised (round 6 replayed a deliberately overflowing batch, found four places where that
        did not hold and fixed them: tools/dbg_overflow.py, tests/test_gpu_graph.py).  The logits of such a replay are garbage
        by construction: the caller runs the eager forward for that batch (or sizes ``headroom`` so that the flag never rises)."""
        from . import ops
        n = data.pos.size(0)
        if n + 1 > self.capacity:
            raise ValueError("the graph was captured for at most %d points (%d given)" % (self.capacity - 1, n))
        b = int(getattr(data, "num_clouds", None) or getattr(data, "num_graphs", None) or (int(data.batch[-1]) + 1))
        if b + 1 != self.data.num_clouds:
            raise ValueError("the graph was captured for %d clouds (%d given)" % (self.data.num_clouds - 1, b))
        pad = self.capacity - n
        self.n = n
        self.data.pos[:n].copy_(data.pos)
        self.data.pos[n].copy_(data.pos[-1])
        if pad > 1:
            self.data.pos[n + 1:].copy_(self._phantom_positions(data.pos[-1], pad - 1, data.pos.device, data.pos.dtype))
        self.data.batch[:n].copy_(data.batch)
        self.data.batch[n:].fill_(b)
        self.data.curve_idxs[:n].copy_(data.curve_idxs)
        self.data.curve_idxs[n:].zero_()
        if getattr(data, "x", None) is not None:
            self.data.x[:n].copy_(data.x)
            self.data.x[n:].zero_()
        if hasattr(data, "labels") and hasattr(self.data, "labels"):      # (per-cloud category ids of the object sets; the phantom cloud keeps its own)
            self.data.labels[:b].copy_(data.labels)
        if verify:
            try:
                with ops.counts_scope(self.bounds), torch.no_grad(), _generator_state(self._rng):
                    self.bounds.rewind(verifying=True)
                    self.model(self.data, **self.kwargs)
            except ops.CountBounds.Exceeded as e:
                raise self.CapacityExceeded(str(e)) from None
            finally:
                self.bounds.rewind()

    def bounded_eager(self):
        """The captured computation launched kernel by kernel (bounded counts, no read-back)."""
        from . import ops
        with ops.counts_scope(self.bounds), torch.no_grad(), _generator_state(self._rng):
            self.bounds.rewind()
            return self.model(self.data, **self.kwargs)[: self.n]

    def eager(self):
        """The ordinary forward over the same batch (host read-back per count), for A/B timing."""
        with torch.no_grad(), _generator_state(self._rng):
            return self.model(self.data, **self.kwargs)[: self.n]

    def replay(self, check=True):
        """One graph launch; the logits of the real points.  ``check``: read the overflow flag back (the forward's one
        synchronisation) and raise ``CapacityExceeded`` if a count did not fit."""
        self.graph.replay()
        if check and int(self.bounds.overflow.item()) != 0:
            raise self.CapacityExceeded("a data-dependent count exceeded the capacity this graph was captured with "
                                        "(or the batch is not sorted): run the eager forward, or capture again")
        return self.out[: self.n]
