"""Data parallelism over the 8 GPUs of one node: one process per GPU, whole clouds sharded across
ranks, and ONE kind of collective -- a bucketed all-reduce (RCCL over xGMI; backend "nccl" is RCCL
on ROCm) of the gradients, launched from autograd hooks so it overlaps the rest of backward.

The reference has no distributed code at all; this is the new functionality of
section 8(e).  BatchNorm statistics stay per rank (a rank's result equals the reference run on
that rank's sub-batch).  Works with the gloo backend on CPU tensors too (used by the tests).
"""
import contextlib
import os

import torch
import torch.distributed as dist

SLOT_ALIGN = 4          # floats: every parameter's slot in a bucket starts 16-byte aligned (vector / LDS-DMA loads)


def _join_wgrad():
    from .ops import join_wgrad         # (ops imports torch only; no cycle)
    join_wgrad()


def init_process_group_from_env(backend=None):
    """torchrun-style rendezvous (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_initialized():
        # backward, geometry, weight-gradient and RCCL streams: more than the runtime's default 4 hardware queues, and two
        # streams on one queue serialise (measured: the next batch's geometry waited behind the whole backward pass).  Only
        # effective before the HIP runtime initialises -- export it in the launcher otherwise.
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    if world == 1 and os.environ.get("CCN_SINGLE_RANK_GROUP") and not dist.is_initialized():
        # diagnostic (CCN_SINGLE_RANK_GROUP=nccl|gloo): a one-rank process group, so that the data-parallel code path
        # (hooks, joins, one all_reduce call per bucket) can be timed by a single process that has the GPU to itself
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend=os.environ["CCN_SINGLE_RANK_GROUP"], rank=0, world_size=1)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


def shard_clouds(cloud_ids, rank, world):
    """Whole clouds are the unit of work: rank r takes every world-th cloud (independent objects,
    no data-path collective)."""
    return list(cloud_ids)[rank::world]


def bucket_layout(plist):
    """Offsets (in elements) of the parameters of one bucket and the bucket length.  Every slot is rounded up to a
    multiple of SLOT_ALIGN floats: a single odd-sized tensor (a 55-class bias) would otherwise leave every weight behind
    it at a non-16-byte address, off the GEMMs' vector / LDS-DMA paths (and rejected by the bf16 kernels)."""
    offs, off = [], 0
    for p in plist:
        offs.append(off)
        off += (p.numel() + SLOT_ALIGN - 1) // SLOT_ALIGN * SLOT_ALIGN
    return offs, off


class GradientAllReduce:
    """Flat fp32 gradient buckets (~25 MB: a handful of large messages per step so that every xGMI
    link carries traffic and launch latency is amortised), all-reduced as soon as every gradient of
    a bucket has been produced.

    ``param.grad`` tensors are VIEWS into the bucket buffers, so there is no pack/unpack copy; use
    ``zero_grad()`` of this object (or ``set_to_none=False``) between steps.

    Readiness is tracked PER PARAMETER: a bucket is reduced when every parameter of it has reported its gradient
    complete.  Parameters whose gradient autograd accumulates report through the post-accumulate hook (once per
    backward pass); parameters whose gradient the HIP layers add straight into the bucket view report after the LAST
    outstanding use (``note_use`` at forward time, ``use_done`` in backward), so a layer that runs twice before one
    backward pass (two forwards summed into one loss, shared weights) is reduced after both products.
    """

    def __init__(self, module, bucket_bytes=25 * 1024 * 1024, process_group=None):
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # CCN_SINGLE_RANK_GROUP (diagnostic, see init_process_group_from_env): hooks and collectives run with one rank too
        self.reduces = self.world > 1 or (dist.is_initialized() and bool(os.environ.get("CCN_SINGLE_RANK_GROUP")))
        self.params = [p for p in module.parameters() if p.requires_grad]      # registration order (optimizer state)
        self.buckets = []          # (flat buffer, [params], [offsets])
        self._bucket_of = {}
        cur, cur_bytes = [], 0
        for p in reversed(self.params):  # backward produces gradients roughly in reverse registration order
            cur.append(p)
            cur_bytes += p.numel() * p.element_size()
            if cur_bytes >= bucket_bytes:
                self._close(cur)
                cur, cur_bytes = [], 0
        if cur:
            self._close(cur)
        self._ready = [set() for _ in self.buckets]
        self._launched = [False] * len(self.buckets)
        self._uses = {}            # id(param) -> outstanding main-grad uses (forward seen, backward not yet)
        self._fused = set()        # id(param) of the parameters whose layers add the gradient into the bucket themselves
        self._handles = []
        self._quiet = False
        self.reduce_calls = 0      # collectives launched since construction (tests)
        # what a multi-rank run reports about itself (bench.py): bytes handed to all_reduce, buckets that had to be reduced
        # in finish() because not every parameter had reported during backward (no overlap for those), host time finish()
        # spent waiting for the collectives
        self.stats = {"steps": 0, "bytes_reduced": 0, "buckets_reduced_in_finish": 0, "finish_wait_s": 0.0}
        if self.reduces:
            for p in self.params:
                p.register_post_accumulate_grad_hook(self._on_grad)

    def _close(self, plist):
        offs, total = bucket_layout(plist)
        flat = torch.zeros(total, dtype=plist[0].dtype, device=plist[0].device)
        for p, off in zip(plist, offs):
            p.grad = flat[off: off + p.numel()].view_as(p)
            # the HIP layers add a weight gradient straight into this view (ops.LinearBNAct: the product already
            # accumulates) instead of returning a tensor for autograd to add -- one launch per layer instead of three
            # (zero-fill, product, add); they report through note_use / use_done
            p._ccn_main_grad = p.grad
            p._ccn_sync = self
            self._bucket_of[p] = len(self.buckets)
        self.buckets.append((flat, list(plist), offs))

    @contextlib.contextmanager
    def no_sync(self):
        """Gradient accumulation over several backward passes: hooks stay quiet inside the block and
        ``finish()`` reduces every bucket once."""
        self._quiet = True
        try:
            yield
        finally:
            self._quiet = False

    # ---- main-grad protocol of the HIP layers (ops._main_grad_*)
    def note_use(self, p):
        """Forward of a layer that will add this parameter's gradient into the bucket view itself."""
        self._uses[id(p)] = self._uses.get(id(p), 0) + 1
        self._fused.add(id(p))

    def use_done(self, p):
        """Backward of such a layer has queued its product; the parameter is complete after its last use."""
        left = self._uses.get(id(p), 0) - 1
        self._uses[id(p)] = max(left, 0)
        if left <= 0:
            self._mark(p, False)

    def use_cancelled(self, p):
        """A noted use is not fused after all; autograd's hook will report the parameter."""
        self._uses[id(p)] = max(self._uses.get(id(p), 0) - 1, 0)

    def _on_grad(self, p):
        """autograd's post-accumulate hook: fires once per backward pass after the parameter's last use, when autograd
        itself accumulated a gradient tensor for it.  (AccumulateGrad skips its hooks for an undefined gradient, so a
        parameter whose layers all added their product into the bucket view and returned None is reported by
        ``use_done`` alone; where both fire in one pass, ``_mark`` recognises the second report.)"""
        if self._uses.get(id(p), 0) > 0:
            return                  # a fused use is still outstanding: use_done reports
        self._mark(p, True)

    def _mark(self, p, from_hook):
        if self._quiet or not self.reduces:
            return
        b = self._bucket_of[p]
        if id(p) in self._ready[b]:
            if from_hook and id(p) in self._fused:
                return              # the same pass, reported by use_done already
            raise RuntimeError("a gradient arrived after its parameter had been reported complete in this step: wrap "
                               "every backward pass but the last in no_sync() when accumulating over several passes")
        self._ready[b].add(id(p))
        if len(self._ready[b]) == len(self.buckets[b][1]):
            self._reduce(b)

    def _reduce(self, b):
        if self._launched[b]:
            raise RuntimeError("gradient bucket %d would be all-reduced twice in one step" % b)
        self._launched[b] = True
        self.reduce_calls += 1
        flat = self.buckets[b][0]
        self.stats["bytes_reduced"] += flat.numel() * flat.element_size()
        if os.environ.get("CCN_DP_DRYRUN"):       # diagnostic: hooks and bookkeeping without the collective itself
            return
        from .ops import wgrad_stream_of
        ws = wgrad_stream_of(flat.device) if flat.is_cuda else None
        if ws is None:
            _join_wgrad()           # (no side stream in use, or CPU tensors: plain ordering on the current stream)
            self._handles.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            return
        # The bucket's weight gradients are still being added on the weight-gradient stream.  Joining that stream into
        # the backward stream here would serialise the rest of backward behind them (and with it the overlap the side
        # stream exists for): instead the COLLECTIVE is ordered behind both -- it is launched from the side stream
        # (RCCL orders itself behind the stream it is called on), after that stream has waited for what backward has
        # queued so far (the BatchNorm / bias gradients of the bucket).
        ws.wait_stream(torch.cuda.current_stream(flat.device))
        with torch.cuda.stream(ws):
            self._handles.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        """Call after ``loss.backward()``: waits for the collectives and averages over ranks."""
        _join_wgrad()
        if self.reduces:
            # buckets whose parameters did not all report a gradient this step are reduced here
            import time
            for b in range(len(self.buckets)):
                if not self._launched[b]:
                    self.stats["buckets_reduced_in_finish"] += 1
                    self._reduce(b)
            t0 = time.perf_counter()
            for h in self._handles:
                h.wait()
            self.stats["finish_wait_s"] += time.perf_counter() - t0
            self.stats["steps"] += 1
            for flat, _, _ in self.buckets:
                flat.div_(self.world)
        self._handles = []
        self._ready = [set() for _ in self.buckets]
        self._launched = [False] * len(self.buckets)
        self._uses.clear()

    def zero_grad(self):
        _join_wgrad()
        for flat, plist, offs in self.buckets:
            flat.zero_()
            for p, off in zip(plist, offs):        # re-attach in case an optimizer replaced .grad (set_to_none=True)
                if p.grad is None or p.grad.data_ptr() != flat[off: off + p.numel()].data_ptr():
                    p.grad = flat[off: off + p.numel()].view_as(p)
                    p._ccn_main_grad = p.grad
        self._ready = [set() for _ in self.buckets]
        self._launched = [False] * len(self.buckets)
        self._uses.clear()

    @property
    def num_bytes(self):
        return sum(f.numel() * f.element_size() for f, _, _ in self.buckets)


class FlatAdam(torch.optim.Optimizer):
    """``torch.optim.Adam`` (the reference harness' optimiser, src/main.py:56) over the flat buckets of a
    ``GradientAllReduce``: the parameters of a bucket are re-homed into one contiguous buffer laid out like
    the gradient bucket, and a step is one ``ccn_adam_step`` launch per bucket (a handful per step instead
    of several hundred per-tensor updates, whose host-side cost left the GPU idle between steps).

    It IS a ``torch.optim.Optimizer``: one ``param_groups`` entry whose ``lr`` every step reads (so the reference's
    ``load_scheduler(config, optimizer)`` -- ExponentialLR / CosineAnnealingWarmRestarts, src/utils/load_utils.py:45-70
    -- drives it unchanged), and ``state_dict()`` / ``load_state_dict()`` use ``torch.optim.Adam``'s format (per
    parameter ``step`` / ``exp_avg`` / ``exp_avg_sq`` in ``module.parameters()`` order), so ``latest_optimizer.pth``
    preemption checkpoints (src/main.py:131-141) are interchangeable with the reference's and do not depend on the
    bucket size.

    Build it after the module sits on its device; ``module.to(...)`` afterwards would detach the views.
    """

    def __init__(self, sync, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.sync = sync
        defaults = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, amsgrad=False, maximize=False,
                        foreach=None, capturable=False, differentiable=False, fused=None)
        self.steps = 0
        self.flat = []             # (flat params, exp_avg, exp_avg_sq) per bucket
        for gflat, plist, offs in sync.buckets:
            if not gflat.is_cuda:
                raise RuntimeError("FlatAdam runs the HIP kernel: parameters must be on the GPU")
            pflat = torch.zeros_like(gflat)
            with torch.no_grad():
                for p, off in zip(plist, offs):
                    view = pflat[off: off + p.numel()].view_as(p)
                    view.copy_(p)
                    p.data = view
            self.flat.append((pflat, torch.zeros_like(gflat), torch.zeros_like(gflat)))
        super().__init__(list(sync.params), defaults)

    @property
    def lr(self):
        return self.param_groups[0]["lr"]

    def zero_grad(self, set_to_none=False):
        self.sync.zero_grad()

    @torch.no_grad()
    def step(self, closure=None):
        from ._lib import call, ptr
        loss = closure() if closure is not None else None
        _join_wgrad()
        self.steps += 1
        g = self.param_groups[0]
        for (gflat, _, _), (pflat, m, v) in zip(self.sync.buckets, self.flat):
            call("adam_step", ptr(pflat), ptr(gflat), ptr(m), ptr(v), gflat.numel(), float(g["lr"]), float(g["betas"][0]),
                 float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]), self.steps)
        return loss

    def _moment_views(self):
        """parameter -> (exp_avg view, exp_avg_sq view)"""
        out = {}
        for (_, plist, offs), (_, m, v) in zip(self.sync.buckets, self.flat):
            for p, off in zip(plist, offs):
                out[p] = (m[off: off + p.numel()].view_as(p), v[off: off + p.numel()].view_as(p))
        return out

    def state_dict(self):
        views = self._moment_views()
        state = {}
        for i, p in enumerate(self.sync.params):
            m, v = views[p]
            state[i] = {"step": torch.tensor(float(self.steps)), "exp_avg": m.clone(), "exp_avg_sq": v.clone()}
        group = {k: v for k, v in self.param_groups[0].items() if k != "params"}
        group["params"] = list(range(len(self.sync.params)))
        return {"state": state if self.steps > 0 else {}, "param_groups": [group]}

    def load_state_dict(self, sd):
        views = self._moment_views()
        group = sd["param_groups"][0]
        if len(group["params"]) != len(self.sync.params):
            raise ValueError("loaded state dict has a different number of parameters")
        for k, v in group.items():
            if k != "params":
                self.param_groups[0][k] = v
        steps = 0
        for i, p in enumerate(self.sync.params):
            st = sd["state"].get(i)
            m, v = views[p]
            if st is None:
                m.zero_()
                v.zero_()
                continue
            m.copy_(st["exp_avg"])
            v.copy_(st["exp_avg_sq"])
            steps = max(steps, int(float(st["step"])))
        self.steps = steps
