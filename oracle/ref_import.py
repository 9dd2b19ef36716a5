"""Loader for the upstream reference (TEST INFRASTRUCTURE ONLY -- never imported by the product).

Only usable inside the build container where ``/root/reference`` is mounted; the GPU box
never sees it.  The reference's hot-path files import third-party packages that are absent
here (pytorch3d, torch_scatter, torch_sparse, torch_cluster, torch_geometric, frnn -- see SURVEY.md
section 8c / App. D).  This module registers stand-in modules for those *imports* so that the
reference's OWN Python -- ``fast_conv1d.py``, ``point_ops.py``, ``fps_ops.py``, and (round 4)
``base.py`` ``ModelBase``, ``pointnet2.py``, ``point_conv.py``, ``dgcnn.py``, ``mlp.py``,
``skip_connect.py`` -- imports in place and runs on CPU.  Nothing from the reference is copied: it is
imported where it lies and only its *outputs* are saved as golden vectors by ``oracle/gen_golden.py``.

Every stand-in below is THE BUILDER'S RESTATEMENT OF A THIRD-PARTY PRIMITIVE (never of reference
code), written from the published semantics of the pinned packages (``setup.sh:13-29``):

    torch_scatter 2.1.1   scatter_add / scatter_mean / scatter_max / scatter_min
    PyG 2.3.0             nn.MLP, nn.conv.MessagePassing (``propagate`` of a bipartite edge list:
                          x_j = x[0][ei[0]], pos_i = pos[1][ei[1]], aggregate over ei[1]),
                          nn.conv.point_conv.PointNetConv (constructor only), nn.inits.reset,
                          utils.softmax, nn.glob.global_add_pool, data.batch.Batch (name only)
    FRNN                  frnn_gather; ``point_ops.fast_knn`` (the reference's CUDA-only wrapper of
                          ``frnn.frnn_grid_points``, point_ops.py:431-461) is rebound to the exhaustive
                          fixed-radius search of ``oracle/frnn_bruteforce.c``
    pytorch3d             ops.knn_points / ball_query (exhaustive C search), sample_farthest_points

What a golden generated through these shims pins: the reference-authored glue -- ``base.py:16-215``,
``pointnet2.py:33-205``, ``point_conv.py:12-93``, ``dgcnn.py:130-266``, ``mlp.py``, ``skip_connect.py`` --
bit for bit.  What it cannot pin: the third-party primitives themselves (listed in DESIGN.md section 2).
"""
import inspect
import os
import sys
import types

import torch
import torch.nn.functional as F

REFERENCE_ROOT = os.environ.get("CCN_REFERENCE_ROOT", "/root/reference")


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "src", "models"))


# ------------------------------------------------------------------------------------------
# torch_scatter 2.1.1 (restated; third-party)
# ------------------------------------------------------------------------------------------


def _dim0(src, dim):
    dim = dim + src.dim() if dim < 0 else dim
    assert dim == 0, "the reference scatters along the row dimension only"


def _segments(index, dim_size):
    if dim_size is None:
        dim_size = int(index.max().item()) + 1 if index.numel() else 0
    return dim_size


def _scatter_add(src, index, dim=0, out=None, dim_size=None):
    _dim0(src, dim)
    res = torch.zeros((_segments(index, dim_size),) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    return res.index_add_(0, index, src)


def _global_add_pool(x, batch, size=None):
    return _scatter_add(x, batch, dim=0, dim_size=size)


def _scatter_mean(src, index, dim=0, out=None, dim_size=None):
    _dim0(src, dim)
    n = _segments(index, dim_size)
    count = torch.zeros(n, dtype=src.dtype).index_add_(0, index, torch.ones(index.numel(), dtype=src.dtype))
    return _scatter_add(src, index, 0, None, n) / count.clamp(min=1).view((-1,) + (1,) * (src.dim() - 1))


def _scatter_min(src, index, dim=0, out=None, dim_size=None):
    assert dim == 0
    if dim_size is None:
        dim_size = int(index.max().item()) + 1
    vals = torch.full((dim_size,) + tuple(src.shape[1:]), float("inf"), dtype=src.dtype)
    arg = torch.full((dim_size,) + tuple(src.shape[1:]), src.shape[0], dtype=torch.long)
    for i in range(src.shape[0]):  # tiny fixtures only
        s = int(index[i])
        better = src[i] < vals[s]
        vals[s] = torch.where(better, src[i], vals[s])
        arg[s] = torch.where(better, torch.full_like(arg[s], i), arg[s])
    return vals, arg


def _scatter_max(src, index, dim=0, out=None, dim_size=None):
    """(values, argmax): per segment and column the FIRST row attaining the maximum; empty segments give 0 and
    arg = src.size(0).  The gradient goes to that single row (torch_scatter's backward is a gather on arg)."""
    _dim0(src, dim)
    n, e = _segments(index, dim_size), src.size(0)
    wide = index.view((-1,) + (1,) * (src.dim() - 1)).expand_as(src)
    with torch.no_grad():
        top = torch.full((n,) + tuple(src.shape[1:]), float("-inf"), dtype=src.dtype)
        top = top.scatter_reduce(0, wide, src, "amax", include_self=True)
        rows = torch.arange(e).view((-1,) + (1,) * (src.dim() - 1)).expand_as(src)
        cand = torch.where(src == top.index_select(0, index), rows, torch.full_like(rows, e))
        arg = torch.full((n,) + tuple(src.shape[1:]), e, dtype=torch.long).scatter_reduce(0, wide, cand, "amin", include_self=True)
    filled = arg < e
    vals = torch.where(filled, src.gather(0, arg.clamp(max=max(e - 1, 0))), torch.zeros((), dtype=src.dtype))
    return vals, arg


# ------------------------------------------------------------------------------------------
# PyG 2.3.0 (restated; third-party)
# ------------------------------------------------------------------------------------------


def _reset(value):
    """torch_geometric.nn.inits.reset"""
    if hasattr(value, "reset_parameters"):
        value.reset_parameters()
    else:
        for child in value.children() if hasattr(value, "children") else []:
            _reset(child)


def _pyg_softmax(src, index, ptr=None, num_nodes=None, dim=0):
    """torch_geometric.utils.softmax: exp(src - segment max) / (segment sum + 1e-16), per column."""
    _dim0(src, dim)
    n = int(index.max().item()) + 1 if num_nodes is None else num_nodes
    top = _scatter_max(src.detach(), index, 0, None, n)[0]
    out = (src - top.index_select(0, index)).exp()
    total = _scatter_add(out, index, 0, None, n) + 1e-16
    return out / total.index_select(0, index)


class _Batch:                                       # torch_geometric.data.batch.Batch: only named by base.py:134-135
    @staticmethod
    def from_data_list(items):
        raise NotImplementedError("data-parallel list input is never exercised by the reference (SURVEY.md section 0)")


class _BatchNorm(torch.nn.Module):
    """torch_geometric.nn.norm.BatchNorm: the real BatchNorm1d lives in ``.module``."""

    def __init__(self, channels):
        super().__init__()
        self.module = torch.nn.BatchNorm1d(channels)

    def reset_parameters(self):
        self.module.reset_parameters()

    def forward(self, x):
        return self.module(x)


class _MLP(torch.nn.Module):
    """torch_geometric.nn.MLP as the reference uses it: per hidden layer Linear -> BatchNorm -> act ->
    dropout; with ``plain_last`` the last layer is Linear (-> dropout with p forced to 0)."""

    def __init__(self, channel_list, dropout=0.0, act="relu", norm="batch_norm", plain_last=True, bias=True, **kwargs):
        super().__init__()
        assert norm == "batch_norm" and act in ("relu", "leaky_relu") and not kwargs.get("act_first", False)
        self.channel_list, self.plain_last = list(channel_list), plain_last
        layers = len(channel_list) - 1
        self.dropout = [float(dropout)] * layers if isinstance(dropout, (int, float)) else [float(d) for d in dropout]
        if isinstance(dropout, (int, float)) and plain_last and layers:
            self.dropout[-1] = 0.0
        self.act = torch.nn.ReLU() if act == "relu" else torch.nn.LeakyReLU()
        self.lins = torch.nn.ModuleList(torch.nn.Linear(a, b, bias=bias) for a, b in zip(channel_list[:-1], channel_list[1:]))
        self.norms = torch.nn.ModuleList(_BatchNorm(c) for c in (channel_list[1:-1] if plain_last else channel_list[1:]))

    def reset_parameters(self):
        for m in list(self.lins) + list(self.norms):
            m.reset_parameters()

    def forward(self, x):
        for i, (lin, norm) in enumerate(zip(self.lins, self.norms)):
            x = F.dropout(self.act(norm(lin(x))), p=self.dropout[i], training=self.training)
        if self.plain_last:
            x = F.dropout(self.lins[-1](x), p=self.dropout[-1], training=self.training)
        return x


class _MessagePassing(torch.nn.Module):
    """torch_geometric.nn.conv.MessagePassing, flow source_to_target: ``propagate`` collects the ``*_j`` arguments of
    ``message`` from side 0 of a (tuple) input at ``edge_index[0]`` and the ``*_i`` arguments from side 1 at
    ``edge_index[1]``, then ``aggregate(messages, edge_index[1], dim_size = #destination nodes)`` and ``update`` (identity)."""

    def __init__(self, aggr="add", *, aggr_kwargs=None, flow="source_to_target", node_dim=-2, decomposed_layers=1, **kwargs):
        super().__init__()
        assert flow == "source_to_target"
        self.aggr, self.flow, self.node_dim = aggr, flow, node_dim

    def reset_parameters(self):
        pass

    def propagate(self, edge_index, size=None, **kwargs):
        sizes, picked = [None, None], {}
        for name in inspect.signature(self.message).parameters:
            side = {"_j": 0, "_i": 1}[name[-2:]]
            value = kwargs[name[:-2]]
            if isinstance(value, (tuple, list)):
                assert len(value) == 2
                if torch.is_tensor(value[1 - side]):
                    sizes[1 - side] = value[1 - side].size(0)
                value = value[side]
            elif torch.is_tensor(value):
                sizes[1 - side] = value.size(0)
            if torch.is_tensor(value):
                sizes[side] = value.size(0)
                value = value.index_select(0, edge_index[side])
            picked[name] = value
        out = self.message(**picked)
        return self.update(self.aggregate(out, edge_index[1], ptr=None, dim_size=sizes[1]))

    def update(self, inputs):
        return inputs


class _PointNetConv(_MessagePassing):
    """torch_geometric.nn.conv.point_conv.PointNetConv: the reference subclasses it and overrides forward / message /
    aggregate (point_conv.py:12-93); what is left of the base class is its constructor."""

    def __init__(self, local_nn=None, global_nn=None, add_self_loops=True, **kwargs):
        kwargs.setdefault("aggr", "max")
        super().__init__(**kwargs)
        self.local_nn, self.global_nn, self.add_self_loops = local_nn, global_nn, add_self_loops
        self.reset_parameters()

    def reset_parameters(self):
        for m in (self.local_nn, self.global_nn):
            if m is not None:
                _reset(m)


def _unused(*a, **k):
    raise NotImplementedError("third-party function on a path no shipped config takes")


# ------------------------------------------------------------------------------------------
# FRNN / pytorch3d (restated; third-party): exhaustive searches of oracle/frnn_bruteforce.c
# ------------------------------------------------------------------------------------------


def _frnn_gather(x, idxs, lengths):
    """frnn.frnn_gather (App. C): out[b, i, k] = x[b, idxs[b, i, k]], zero rows where idxs < 0; differentiable in x."""
    b, p1, k = idxs.shape
    flat = idxs.clamp(min=0).reshape(b, p1 * k, 1).expand(-1, -1, x.size(-1))
    out = x.gather(1, flat).view(b, p1, k, x.size(-1))
    return out * (idxs >= 0).unsqueeze(-1).to(x.dtype)


def _fast_knn(points1, points2, lengths1, lengths2, K, r, return_nn=False):
    """Rebinds the reference's ``fast_knn`` (point_ops.py:431-461: CUDA-only argument checks + ``frnn.frnn_grid_points``,
    of which it keeps ``idxs`` alone) to the exhaustive fixed-radius search."""
    from oracle import torch_ref as R
    if points1.shape[0] != points2.shape[0]:
        raise ValueError("points1 and points2 must have the same batch  dimension")
    return R.frnn_bruteforce(points1.contiguous(), points2.contiguous(), lengths1, lengths2, K, r)


def _knn_points(p1, p2, lengths1=None, lengths2=None, K=1, return_nn=False, **kwargs):
    from oracle import torch_ref as R
    return None, R.knn_bruteforce(p1, p2, lengths1, lengths2, K), None


def _ball_query(p1, p2, lengths1=None, lengths2=None, K=500, radius=0.2, return_nn=True):
    from oracle import torch_ref as R
    query = R.ball_query_bruteforce if p1.size(2) == 3 else R.ball_query_nd
    return None, query(p1, p2, lengths1, lengths2, K, radius), None


def _sample_farthest_points(points, lengths=None, K=50, random_start_point=False):
    """pytorch3d.ops.sample_farthest_points with a per-cloud K tensor: (None, idx (B, max K) padded with -1).  The start
    of cloud b is one ``torch.randint(lengths[b], (1,))`` draw, taken cloud by cloud (the draw is what a test injects)."""
    keep = [int(k) for k in torch.as_tensor(K).reshape(-1).tolist()]
    keep = keep * points.size(0) if len(keep) == 1 else keep
    idx = torch.full((points.size(0), max(keep)), -1, dtype=torch.long)
    for b in range(points.size(0)):
        n = int(lengths[b])
        p = points[b, :n]
        cur = int(torch.randint(n, (1,))) if random_start_point else 0
        dist = torch.full((n,), float("inf"), dtype=points.dtype)
        for s in range(min(keep[b], n)):
            idx[b, s] = cur
            d = p - p[cur]
            dist = torch.minimum(dist, (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2])
            cur = int(torch.argmax(dist))
    return None, idx


def _stub(name, **attrs):
    mod = types.ModuleType(name)
    mod.__dict__.update(attrs)
    sys.modules[name] = mod
    return mod


_INSTALLED = False


def _install_stubs():
    global _INSTALLED
    if _INSTALLED:
        return
    if not reference_available():
        raise RuntimeError("reference tree not present (expected only in the build container)")
    sys.dont_write_bytecode = True  # never write __pycache__ into the read-only reference
    _stub("pytorch3d")
    _stub("pytorch3d.ops", sample_farthest_points=_sample_farthest_points, ball_query=_ball_query, knn_points=_knn_points)
    _stub("torch_scatter", scatter_add=_scatter_add, scatter_max=_scatter_max, scatter_mean=_scatter_mean,
          scatter_min=_scatter_min)
    _stub("torch_sparse", SparseTensor=type("SparseTensor", (), {}), set_diag=_unused)
    _stub("torch_cluster", knn=_unused)
    _stub("torch_geometric")
    _stub("torch_geometric.nn", knn=_unused, MLP=_MLP, fps=_unused, radius=_unused, global_mean_pool=_unused)
    _stub("torch_geometric.nn.glob", global_add_pool=_global_add_pool)
    _stub("torch_geometric.nn.conv", MessagePassing=_MessagePassing)
    _stub("torch_geometric.nn.conv.point_conv", PointNetConv=_PointNetConv)
    _stub("torch_geometric.nn.inits", reset=_reset)
    _stub("torch_geometric.utils", softmax=_pyg_softmax, add_self_loops=_unused, remove_self_loops=_unused)
    _stub("torch_geometric.typing", OptTensor=None, Adj=None, PairOptTensor=None, PairTensor=None)
    _stub("torch_geometric.data")
    _stub("torch_geometric.data.batch", Batch=_Batch)
    _stub("frnn", frnn_gather=_frnn_gather, frnn_grid_points=_unused)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    _INSTALLED = True


def load_reference():
    """Returns (fast_conv1d, point_ops, fps_ops) modules of the reference, imported in place."""
    _install_stubs()
    import src.models.modules.fast_conv1d as fc
    import src.models.utils.point_ops as po
    import src.models.modules.fps_ops as fo
    return fc, po, fo


def load_reference_model():
    """The reference's model assembly imported in place: a namespace with ``base`` (``ModelBase``), ``pointnet2``,
    ``point_conv``, ``dgcnn``, ``mlp``, ``skip_connect``, ``fast_conv1d``, ``point_ops``, ``fps_ops`` -- its own code over the
    third-party stand-ins above."""
    fc, po, fo = load_reference()
    po.fast_knn = _fast_knn                          # (CUDA-only wrapper of the absent FRNN package, see _fast_knn)
    import src.models.base as base
    import src.models.modules.pointnet2 as pointnet2
    import src.models.modules.point_conv as point_conv
    import src.models.modules.dgcnn as dgcnn
    import src.models.modules.mlp as mlp
    import src.models.modules.skip_connect as skip_connect
    return types.SimpleNamespace(base=base, pointnet2=pointnet2, point_conv=point_conv, dgcnn=dgcnn, mlp=mlp,
                                 skip_connect=skip_connect, fast_conv1d=fc, point_ops=po, fps_ops=fo, MLP=_MLP)


def load_reference_harness():
    """Returns (lovasz_losses module, SemKITTI, SemNuScenes) of the reference, imported in place: the Lovasz loss of
    the training harness and the dataset-side curve splitters (only their ``_get_curves`` methods are called)."""
    load_reference()

    class _Base:                                   # torch_geometric.data.Dataset / Data are only base classes here
        def __init__(self, *a, **k):
            pass

    sys.modules["torch_geometric.data"].__dict__.update(Data=_Base, Dataset=_Base)
    for name in ("cv2", "mitsuba", "nuscenes"):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:                       # noqa: BLE001 -- visualisation-only dependencies
                _stub(name)
    _stub("src.visualization.mitsuba_render", __getattr__=lambda name: None)      # render_pc_* helpers: never called
    import src.models.utils.lovasz_losses as lov
    import src.data.kitti_dataset as kd
    import src.data.nuscenes_dataset as nd
    return lov, kd.SemKITTI, nd.SemNuScenes


def load_reference_runners():
    """The reference's task runners imported in place for their LOSS functions only:
    ``kitti_seg.seg_loss_kitti`` (kitti_seg.py:184-200), ``nuscenes_seg.seg_loss`` (:229-233), ``audi_seg.seg_loss_audi``
    (:178-182).  wandb / torchmetrics / cv2 / mitsuba are logging and visualisation imports: empty stand-ins."""
    load_reference_harness()
    for name in ("wandb", "torchmetrics", "tqdm"):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:                       # noqa: BLE001
                _stub(name)
    import src.run.kitti_seg as kitti_seg
    import src.run.nuscenes_seg as nuscenes_seg
    import src.run.audi_seg as audi_seg
    return kitti_seg, nuscenes_seg, audi_seg
