"""``ModelBase``: assembles the step modules from the reference's config dict (same ``model:``
section of the YAMLs, same state-dict keys) and runs its skip-connection state machine
(ref src/models/base.py:16-215).
"""
import copy

import torch
import torch.nn.functional as F
import yaml

from .nn import MLP
from .steps import (CurveFPModule, CurveSAModule, DGCNNLayer, DGCNNLayerRadius, ForwardContext, FPModule, GlobalSAModule,
                    SAModule, SGCNNLayer, SharedMLP,
                    SkipConnect, SymmetricCurve1DConvFastV1, SymmetricCurve1DConvV2)

_DOWNSAMPLING_STEPS = ("sa", "sa-geo", "sa-global", "pt-transition-down")
_FEATURE_ONLY_STEPS = ("mlp", "skip-connect")


_PREPARE_POOL = None            # one worker thread for ModelBase.prepare_async


class ModelBase(torch.nn.Module):
    def __init__(self, in_dim, n_out, steps=("conv1d", "dgcnn", "conv1d", "sa", "sa", "sa-global"),
                 feat_dims=((32, 32, 64), (64, 128), (128, 128), (128, 128, 256), (256, 512, 1024), (1024,)),
                 out_mlp=dict(), **kwargs):
        super().__init__()
        self.in_dim, self.n_out = in_dim, n_out
        self.use_bias = kwargs.get("use_bias", False)
        self.version = kwargs.get("version", 2.0)
        self.mlp_func = MLP
        self.step_names = list(steps)
        self.skip_connect_state_store = list(kwargs.get("skip_connect_state_store", []))

        self.steps = torch.nn.ModuleList()
        for i, step_name in enumerate(steps):
            step_kwargs = kwargs.copy()
            if isinstance(step_name, dict):
                step_kwargs = {**step_kwargs, **step_name}
                step_name = step_kwargs.pop("step_name")
                self.step_names[i] = step_name
            step_kwargs["with_xyz"] = step_kwargs.get("with_xyz", False)
            dims = self._get_input_dim(i, step_name, feat_dims, in_dim, step_kwargs["with_xyz"])
            self.steps.append(self.add_step(i, step_name, dims, **step_kwargs))

        final_mlp_kwargs = {"dropout": 0.5, "norm": "batch_norm", "plain_last": True}
        spec = copy.deepcopy(out_mlp)
        if isinstance(spec, dict):
            hidden = spec.pop("dims", None) or []
            final_mlp_kwargs.update(spec)
        else:
            hidden = spec
        dims = [feat_dims[-1][-1]] + list(hidden) + [n_out]
        if final_mlp_kwargs.pop("with_seg_category", False):
            dims[0] += 64
            self.lin_categorical = self.mlp_func([16, 64, 64])
        if final_mlp_kwargs.pop("identity", False):
            self.mlp = torch.nn.Identity()
        else:
            self.mlp = self.mlp_func(dims, bias=self.use_bias, **final_mlp_kwargs)

    # ---- ref base.py:66-84
    def _get_input_dim(self, step_idx, step_name, feat_dims, in_dim, with_xyz):
        prev = in_dim if step_idx == 0 else feat_dims[step_idx - 1][-1]
        if step_name in ("dgcnn", "sgcnn"):
            head = [in_dim * 2] if step_idx == 0 else [2 * (prev + 3 * with_xyz)]
        elif step_name in ("sa", "sa-global", "sa-geo"):
            head = [in_dim + 3 * with_xyz] if step_idx == 0 else [prev + 3 + 3 * with_xyz]
        elif step_idx == 0:
            head = [in_dim]
        elif step_name in ("skip-connect", "fp", "fp-geo"):
            head = []
        elif step_name in ("mlp", "conv1d-fast-v1", "conv1d-fast-v2"):
            head = [prev + 3 * with_xyz]
        else:
            raise NotImplementedError("No Module Named >> %s" % step_name)
        return head + list(feat_dims[step_idx])

    def _attend_nn(self, dims, kwargs, halve_v2):
        if kwargs.get("aggr_type") not in ("attend", "weighted-sum"):
            return None
        c = dims[-1]
        mid = c // 2 if (halve_v2 and self.version == 2.0) else c
        return self.mlp_func([c, mid, c], act="leaky_relu", bias=self.use_bias)

    # ---- ref base.py:86-131
    def add_step(self, step_idx, step_name, dims, **kwargs):
        b = self.use_bias
        if step_name == "sa":
            return SAModule(kwargs["ratios"][step_idx], kwargs["radii"][step_idx], self.mlp_func(dims, bias=b),
                            attend_nn=self._attend_nn(dims, kwargs, True), k=kwargs["knn"][step_idx], **kwargs)
        if step_name == "sgcnn":
            return SGCNNLayer(self.mlp_func(dims, bias=b), kwargs["knn"][step_idx], r=kwargs["radii"][step_idx],
                              attend_nn=self._attend_nn(dims, kwargs, False), **kwargs)
        if step_name == "sa-geo":
            return CurveSAModule(kwargs["ratios"][step_idx], kwargs["radii"][step_idx],
                                 self.mlp_func(dims, act="leaky_relu", bias=b),
                                 attend_nn=self._attend_nn(dims, kwargs, False), **kwargs)
        if step_name == "conv1d-fast-v1":
            return SymmetricCurve1DConvFastV1(dims, kwargs["kernel_sizes"][step_idx], with_xyz=kwargs["with_xyz"],
                                              with_diff=kwargs.get("with_diff", False))
        if step_name == "conv1d-fast-v2":
            return SymmetricCurve1DConvV2(dims, kwargs["kernel_sizes"][step_idx], with_xyz=kwargs["with_xyz"],
                                          with_diff=kwargs.get("with_diff", False))
        if step_name == "skip-connect":
            return SkipConnect(self.mlp_func(dims, act="leaky_relu", bias=b), kwargs["num_skips"][step_idx])
        if step_name == "fp":
            return FPModule(kwargs["knn"][step_idx], self.mlp_func(dims, bias=b), with_xyz=kwargs["with_xyz"])
        if step_name == "fp-geo":
            return CurveFPModule(kwargs["knn"][step_idx], self.mlp_func(dims, act="leaky_relu", bias=b),
                                 with_xyz=kwargs["with_xyz"])
        if step_name == "mlp":
            return SharedMLP(dims, **kwargs)
        if step_name == "sa-global":
            return GlobalSAModule(self.mlp_func(dims, bias=b), **kwargs)
        if step_name == "dgcnn":
            return DGCNNLayer(self.mlp_func(dims, bias=b), kwargs["knn"][step_idx], with_xyz=kwargs["with_xyz"])
        if step_name == "dgcnn-rad":
            return DGCNNLayerRadius(self.mlp_func(dims, bias=b), kwargs["radii"][step_idx], with_xyz=kwargs["with_xyz"])
        raise NotImplementedError("Have not implemented step %s yet!" % step_name)

    # ---- ref base.py:133-209
    def prepare(self, data, inputs_ready=True, main_stream=None):
        """Optional pipelining hook for a training loop: computes the position-only part of ``forward(data)`` (sampling,
        neighbour search, index tables of every step) on the side stream NOW -- typically right after
        ``loss.backward()`` of the previous batch has been queued, so that it overlaps that backward pass -- and returns
        a plan to pass as ``forward(data, plan=...)``.  ``inputs_ready``: the tensors of ``data`` are not being produced
        by work still queued on the current stream (true for loader output that was copied earlier).  Returns None
        when there is no side stream or a step searches in feature space; ``forward`` then does everything itself."""
        pos, batch, p2c = data.pos, data.batch, data.curve_idxs
        num_clouds = getattr(data, "num_clouds", None) or getattr(data, "num_graphs", None)
        ctx = ForwardContext(num_clouds, device=pos.device, inputs_ready=inputs_ready, main_stream=main_stream)
        kwargs = {"_ccn_ctx": ctx}
        tables = self._geometry_prepass(ctx, pos, batch, p2c, kwargs)
        return None if tables is None else (ctx, tables, data)

    def prepare_async(self, data, inputs_ready=True, seed=None):
        """``prepare(data)`` on a worker thread: returns a ``concurrent.futures.Future`` whose ``result()`` is the plan.

        The position-only pass reads element counts back between its kernels (sampled points, edges per level: the sizes
        of everything the feature pass allocates), so its HOST time is mostly waiting for the side stream -- 50 ms per
        step on BASELINE configs[4], where exact farthest point sampling over ~17 k-point clouds is a chain of ~6 k
        dependent rounds.  Those waits release the interpreter lock: started at the top of a step, the next batch's
        geometry proceeds while this thread queues the current forward and backward pass.  ``seed``: seeds torch's global CPU
        generator ONLY (``torch.default_generator.manual_seed``), on the worker, before the pass: the sampling draws (CurveFPS
        phase, random / VoxelFPS / FPS starts) come from that generator; the CUDA generators -- the dropout masks of the
        caller's forward pass -- are left alone (``torch.manual_seed`` would reseed them from another thread at an arbitrary
        point of the caller's queueing: ADVICE r4).  The caller's thread must not draw from the CPU generator meanwhile; a
        forward with a plan does not.  The worker shares ``_lib.PROFILE`` with the caller: records of the two threads interleave
        in arrival order (bench.py's per-kernel tables sum per name, so the order does not matter there)."""
        import concurrent.futures
        global _PREPARE_POOL
        if _PREPARE_POOL is None:
            _PREPARE_POOL = concurrent.futures.ThreadPoolExecutor(max_workers=1, thread_name_prefix="ccn-geometry")
        device = data.pos.device
        main = torch.cuda.current_stream(device)          # (the worker thread has a current stream of its own)

        def work():
            torch.cuda.set_device(device)                 # (the current device is per thread as well)
            if seed is not None:
                torch.default_generator.manual_seed(seed)
            return self.prepare(data, inputs_ready=inputs_ready, main_stream=main)
        return _PREPARE_POOL.submit(work)

    def forward(self, data, plan=None, **kwargs):
        x, pos, batch, p2c = data.x, data.pos, data.batch, data.curve_idxs
        if hasattr(data, "labels"):
            kwargs["shapenet-categories"] = data.labels
        num_clouds = getattr(data, "num_clouds", None) or getattr(data, "num_graphs", None)
        if plan is not None:
            if plan[2] is not data:
                raise ValueError("the plan was prepared for another batch")
            ctx, plan = plan[0], plan[1]
            kwargs["_ccn_ctx"] = ctx
        else:
            ctx = kwargs["_ccn_ctx"] = ForwardContext(num_clouds, device=pos.device)
        hist = {"x": [x], "pos": [pos], "batch": [batch], "p2c": [p2c], "idx": []}
        proportional, downsampled = [], []
        cloud_of_point = batch
        for i, (name, step) in enumerate(zip(self.step_names, self.steps)):
            staged = plan[i] if plan is not None else None
            if staged is not None:
                g, ready = staged
                if ready is not None:                       # (None: the consumer synchronised already, graph.CapturedForward)
                    ctx.main.wait_event(ready)              # this step's index tables, produced on the side stream
            if name in ("fp", "fp-geo"):
                j = downsampled.pop()
                x_skip = hist["x"][j] if hist["x"][j] is not None else hist["pos"][j]
                if staged is not None:
                    out = (step.features(x, x_skip, g),) + g.out
                elif name == "fp":
                    out = step(x, pos, batch, x_skip, hist["pos"][j], hist["batch"][j], p2c, hist["p2c"][j], **kwargs)
                else:
                    out = step(x, hist["idx"][j], x_skip, hist["pos"][j], hist["batch"][j], hist["p2c"][j], **kwargs)
            elif name == "skip-connect":
                take = proportional[-step.num_skips:]
                del proportional[-step.num_skips:]
                xs = [x] + [hist["x"][j] if hist["x"][j] is not None else hist["pos"][j] for j in take]
                out = step(xs, pos, batch, p2c, **kwargs)
            elif staged is not None:
                out = (step.features(x, pos, g),) + g.out
            else:
                out = step(x, pos, batch, p2c, **kwargs)
            x, pos, batch, p2c = out[:4]
            hist["x"].append(x)
            hist["pos"].append(pos)
            hist["batch"].append(batch)
            hist["p2c"].append(p2c)
            hist["idx"].append(out[5] if len(out) > 5 else None)
            if name in self.skip_connect_state_store:
                proportional.append(i)
            if name in _DOWNSAMPLING_STEPS:
                downsampled.append(i)
        if "shapenet-categories" in kwargs and hasattr(self, "lin_categorical"):
            cats = F.one_hot(kwargs["shapenet-categories"], num_classes=16).float()
            x = torch.cat([x, self.lin_categorical(cats)[cloud_of_point]], dim=1)
        return self.mlp(x)

    def _geometry_prepass(self, ctx, pos, batch, p2c, kwargs):
        """Sampling, neighbour search and index tables of EVERY step on the side stream (they depend on positions only).
        Inside one forward this would only delay the feature kernels (the host reads element counts back between the
        steps: measured 139 vs 126 ms per step), so ``forward`` interleaves geometry and features step by step; the
        prepass is what ``prepare()`` runs while the previous batch is still in its backward pass.  Returns per step
        (tables, ready-event) or None for the steps without geometry; None altogether when there is no side stream
        or a step searches in feature space (dgcnn).  The CPU generator is drawn from in step order, as in the loop."""
        if ctx.side is None:
            return None
        if any(getattr(s, "geometry_needs_features", False) or not hasattr(s, "geometry") and n not in _FEATURE_ONLY_STEPS
               for n, s in zip(self.step_names, self.steps)):
            return None
        hist = {"pos": [pos], "batch": [batch], "p2c": [p2c], "idx": []}
        downsampled, plan = [], []
        for i, (name, step) in enumerate(zip(self.step_names, self.steps)):
            if name in _FEATURE_ONLY_STEPS:
                plan.append(None)
                out = (pos, batch, p2c)
            else:
                block = ctx.geometry(defer=True)
                with block as geo:
                    if name == "fp":
                        j = downsampled.pop()
                        g = step.geometry(pos, batch, hist["pos"][j], hist["batch"][j], p2c, hist["p2c"][j], kwargs)
                    elif name == "fp-geo":
                        j = downsampled.pop()
                        g = step.geometry(hist["idx"][j], hist["pos"][j], hist["batch"][j], hist["p2c"][j], kwargs)
                    else:
                        g = step.geometry(pos, batch, p2c, kwargs)
                    geo.publish(g)
                plan.append((g, block.event))
                out = g.out
            pos, batch, p2c = out[:3]
            hist["pos"].append(pos)
            hist["batch"].append(batch)
            hist["p2c"].append(p2c)
            hist["idx"].append(out[4] if len(out) > 4 else None)
            if name in _DOWNSAMPLING_STEPS:
                downsampled.append(i)
        return plan


def load_model_config(path):
    """The ``model:`` section of a reference YAML (ref src/utils/load_utils.py:17-27)."""
    with open(path) as f:
        cfg = yaml.safe_load(f)
    return cfg["model"] if "model" in cfg else cfg


def build_model(model_cfg, in_dim, n_out):
    """ref src/utils/load_utils.py:17-27 ``load_model``: ModelBase(in_dim, out_dim, **config['model'])."""
    cfg = {k: v for k, v in copy.deepcopy(model_cfg).items() if k != "type"}
    return ModelBase(in_dim, n_out, **cfg)


def segmentation_loss(logits, target, ignore_index=-100, reduction="mean", check_targets=None):
    """Harness counterpart of ref src/run/kitti_seg.py:184-192: F.nll_loss(F.log_softmax(logits), target) as one fused
    forward and one backward pass (ops.NLLLoss).

    ``reduction="mean"`` (default) divides by the number of rows whose target is not ``ignore_index`` -- torch's
    ``reduction='mean'``, what the nuScenes / A2D2 runners and the benchmark use.  ``reduction="mean_all"`` is the KITTI
    runner's form (kitti_seg.py:184-192: ``nll_loss(reduction='none', ignore_index=0)`` followed by ``torch.mean`` over
    ALL points): ignored rows contribute zero but stay in the denominator.

    A target outside [0, C) that is not ``ignore_index`` contributes nothing in the kernel (torch raises a device-side
    assertion there).  ``check_targets=True`` (or CCN_CHECK_TARGETS=1) validates the range on the host first, at the
    price of one device synchronisation, and raises IndexError like torch's CPU path."""
    import os
    from . import ops
    if reduction not in ("mean", "mean_all"):
        raise ValueError("reduction must be 'mean' or 'mean_all'")
    if check_targets is None:
        check_targets = os.environ.get("CCN_CHECK_TARGETS") == "1"
    if check_targets:
        c = logits.size(-1)
        bad = (target != ignore_index) & ((target < 0) | (target >= c))
        if bool(bad.any()):
            raise IndexError("Target %d is out of bounds." % int(target[bad][0]))
    return ops.NLLLoss.apply(logits, target, ignore_index, reduction == "mean_all")
