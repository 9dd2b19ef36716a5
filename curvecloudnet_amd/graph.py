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
import torch


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
            with torch.cuda.graph(self.graph):
                self.out = model(data, plan=self.plan, **forward_kwargs)
        self.launches = None

    def eager(self):
        """The same pass launched kernel by kernel (for A/B timing and the bit-identity test)."""
        with torch.no_grad():
            return self.model(self.data, plan=self.plan, **self.kwargs)

    def replay(self):
        """One graph launch; returns the (static) logits tensor of the captured pass."""
        self.graph.replay()
        return self.out
