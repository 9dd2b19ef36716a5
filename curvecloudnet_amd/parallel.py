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


def _join_wgrad():
    from .ops import join_wgrad         # (ops imports torch only; no cycle)
    join_wgrad()


def init_process_group_from_env(backend=None):
    """torchrun-style rendezvous (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
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


class GradientAllReduce:
    """Flat fp32 gradient buckets (~25 MB: a handful of large messages per step so that every xGMI
    link carries traffic and launch latency is amortised), all-reduced as soon as every gradient of
    a bucket has been produced.

    ``param.grad`` tensors are VIEWS into the bucket buffers, so there is no pack/unpack copy; use
    ``zero_grad()`` of this object (or ``set_to_none=False``) between steps.
    """

    def __init__(self, module, bucket_bytes=25 * 1024 * 1024, process_group=None):
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # CCN_SINGLE_RANK_GROUP (diagnostic, see init_process_group_from_env): hooks and collectives run with one rank too
        self.reduces = self.world > 1 or (dist.is_initialized() and bool(os.environ.get("CCN_SINGLE_RANK_GROUP")))
        params = [p for p in module.parameters() if p.requires_grad]
        self.buckets = []          # (flat buffer, [params])
        self._bucket_of = {}
        cur, cur_bytes = [], 0
        for p in reversed(params):  # backward produces gradients roughly in reverse registration order
            cur.append(p)
            cur_bytes += p.numel() * p.element_size()
            if cur_bytes >= bucket_bytes:
                self._close(cur)
                cur, cur_bytes = [], 0
        if cur:
            self._close(cur)
        self._pending = [0] * len(self.buckets)
        self._handles = []
        self._quiet = False
        if self.reduces:
            for p in params:
                p.register_post_accumulate_grad_hook(self._on_grad)

    def _close(self, plist):
        flat = torch.zeros(sum(p.numel() for p in plist), dtype=plist[0].dtype, device=plist[0].device)
        off = 0
        for p in plist:
            p.grad = flat[off: off + p.numel()].view_as(p)
            # the HIP layers add a weight gradient straight into this view (ops.LinearBNAct: the product already
            # accumulates with atomics) instead of returning a tensor for autograd to add -- one launch per layer
            # instead of three (zero-fill, product, add); they report completion through _ccn_grad_ready
            p._ccn_main_grad = p.grad
            p._ccn_grad_ready = (lambda q=p: self._on_grad(q)) if self.reduces else None
            off += p.numel()
            self._bucket_of[p] = len(self.buckets)
        self.buckets.append((flat, list(plist)))

    @contextlib.contextmanager
    def no_sync(self):
        """Gradient accumulation over several backward passes: hooks stay quiet inside the block and
        ``finish()`` reduces every bucket once."""
        self._quiet = True
        try:
            yield
        finally:
            self._quiet = False

    def _on_grad(self, p):
        if self._quiet:
            return
        b = self._bucket_of[p]
        self._pending[b] += 1
        if self._pending[b] == len(self.buckets[b][1]):
            _join_wgrad()           # weight gradients of the HIP layers may still be running on their side stream
            self._handles.append(dist.all_reduce(self.buckets[b][0], op=dist.ReduceOp.SUM, group=self.group,
                                                 async_op=True))

    def finish(self):
        """Call after ``loss.backward()``: waits for the collectives and averages over ranks."""
        _join_wgrad()
        if self.reduces:
            # buckets whose parameters did not all receive a gradient this step are reduced here
            for b, (flat, plist) in enumerate(self.buckets):
                if self._pending[b] != len(plist):
                    self._handles.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            for h in self._handles:
                h.wait()
            for flat, _ in self.buckets:
                flat.div_(self.world)
        self._handles = []
        self._pending = [0] * len(self.buckets)

    def zero_grad(self):
        _join_wgrad()
        for flat, plist in self.buckets:
            flat.zero_()
            off = 0
            for p in plist:        # re-attach in case an optimizer replaced .grad
                if p.grad is None or p.grad.data_ptr() != flat[off: off + p.numel()].data_ptr():
                    p.grad = flat[off: off + p.numel()].view_as(p)
                    p._ccn_main_grad = p.grad
                off += p.numel()

    @property
    def num_bytes(self):
        return sum(f.numel() * f.element_size() for f, _ in self.buckets)


class FlatAdam:
    """``torch.optim.Adam`` (the reference harness' optimiser, src/main.py:56) over the flat buckets of a
    ``GradientAllReduce``: the parameters of a bucket are re-homed into one contiguous buffer laid out like
    the gradient bucket, and a step is one ``ccn_adam_step`` launch per bucket (a handful per step instead
    of several hundred per-tensor updates, whose host-side cost left the GPU idle between steps).

    Build it after the module sits on its device; ``module.to(...)`` afterwards would detach the views.
    """

    def __init__(self, sync, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.sync, self.lr, self.betas, self.eps, self.weight_decay = sync, lr, betas, eps, weight_decay
        self.steps = 0
        self.state = []            # (flat params, exp_avg, exp_avg_sq) per bucket
        for gflat, plist in sync.buckets:
            if not gflat.is_cuda:
                raise RuntimeError("FlatAdam runs the HIP kernel: parameters must be on the GPU")
            pflat = torch.empty_like(gflat)
            off = 0
            with torch.no_grad():
                for p in plist:
                    view = pflat[off: off + p.numel()].view_as(p)
                    view.copy_(p)
                    p.data = view
                    off += p.numel()
            self.state.append((pflat, torch.zeros_like(gflat), torch.zeros_like(gflat)))

    def zero_grad(self):
        self.sync.zero_grad()

    @torch.no_grad()
    def step(self):
        from ._lib import call, ptr
        _join_wgrad()
        self.steps += 1
        for (gflat, _), (pflat, m, v) in zip(self.sync.buckets, self.state):
            call("adam_step", ptr(pflat), ptr(gflat), ptr(m), ptr(v), gflat.numel(), float(self.lr), float(self.betas[0]),
                 float(self.betas[1]), float(self.eps), float(self.weight_decay), self.steps)

    def state_dict(self):
        return {"steps": self.steps, "exp_avg": [m for _, m, _ in self.state], "exp_avg_sq": [v for _, _, v in self.state]}

    def load_state_dict(self, sd):
        self.steps = int(sd["steps"])
        for (_, m, v), a, b in zip(self.state, sd["exp_avg"], sd["exp_avg_sq"]):
            m.copy_(a)
            v.copy_(b)
