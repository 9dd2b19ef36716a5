"""Loader for the upstream reference (TEST INFRASTRUCTURE ONLY -- never imported by the product).

Only usable inside the build container where ``/root/reference`` is mounted; the GPU box
never sees it.  The reference's hot-path files import third-party packages that are absent
here (pytorch3d, torch_scatter, torch_geometric, frnn -- see SURVEY.md section 8c / App. D), so
this module registers empty stand-in modules for those *imports* and supplies three small
scatter shims (our own restatement of ``scatter_add`` / ``global_add_pool`` / ``scatter_min``)
so that the reference's pure-PyTorch curve functions run on CPU.  Nothing from the reference
is copied: it is imported in place and only its *outputs* are saved as golden vectors by
``oracle/gen_golden.py``.
"""
import os
import sys
import types

import torch

REFERENCE_ROOT = os.environ.get("CCN_REFERENCE_ROOT", "/root/reference")


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "src", "models"))


def _scatter_add(src, index, dim=0, out=None, dim_size=None):
    assert dim == 0
    if dim_size is None:
        dim_size = int(index.max().item()) + 1 if index.numel() else 0
    res = torch.zeros((dim_size,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    return res.index_add_(0, index, src)


def _global_add_pool(x, batch, size=None):
    return _scatter_add(x, batch, dim=0, dim_size=size)


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


def _stub(name, **attrs):
    mod = types.ModuleType(name)
    mod.__dict__.update(attrs)
    sys.modules[name] = mod
    return mod


def load_reference():
    """Returns (fast_conv1d, point_ops, fps_ops) modules of the reference, imported in place."""
    if not reference_available():
        raise RuntimeError("reference tree not present (expected only in the build container)")
    sys.dont_write_bytecode = True  # never write __pycache__ into the read-only reference
    _stub("pytorch3d")
    _stub("pytorch3d.ops", sample_farthest_points=None, ball_query=None, knn_points=None)
    _stub("torch_scatter", scatter_add=_scatter_add, scatter_max=None, scatter_mean=None,
          scatter_min=_scatter_min)
    _stub("torch_geometric")
    _stub("torch_geometric.nn", knn=None, MLP=None, fps=None, radius=None)
    _stub("torch_geometric.nn.glob", global_add_pool=_global_add_pool)
    _stub("torch_geometric.typing", OptTensor=None, Adj=None, PairOptTensor=None, PairTensor=None)
    _stub("frnn")
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import src.models.modules.fast_conv1d as fc
    import src.models.utils.point_ops as po
    import src.models.modules.fps_ops as fo
    return fc, po, fo


def load_reference_harness():
    """Returns (lovasz_losses module, SemKITTI, SemNuScenes) of the reference, imported in place: the Lovasz loss of
    the training harness and the dataset-side curve splitters (only their ``_get_curves`` methods are called)."""
    load_reference()

    class _Base:                                   # torch_geometric.data.Dataset / Data are only base classes here
        def __init__(self, *a, **k):
            pass

    _stub("torch_geometric.data", Data=_Base, Dataset=_Base)
    for name in ("cv2", "mitsuba", "nuscenes"):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:                       # noqa: BLE001 -- visualisation-only dependencies
                _stub(name)
    _stub("src.visualization.mitsuba_render", render_pc_kitti=None)
    import src.models.utils.lovasz_losses as lov
    import src.data.kitti_dataset as kd
    import src.data.nuscenes_dataset as nd
    return lov, kd.SemKITTI, nd.SemNuScenes
