"""Per-module parity cases of the step modules above the curve ops (TEST INFRASTRUCTURE ONLY).

One table, three users: ``oracle/gen_golden.py`` builds each case from the REFERENCE's own classes
(``oracle/ref_import.load_reference_model``) and stores inputs, draws, ``state_dict``, outputs and gradients in
``tests/golden/modules.npz``; ``tests/test_oracle_golden.py`` builds the same case from the CPU oracle and
``tests/test_gpu_golden.py`` from the HIP product, load the stored ``state_dict`` and compare with the stored results.
The constructor calls are written once against a namespace of class names -- the three sides share the reference's
constructor and forward signatures.

Rows covered: A13 ``PointNetConv2`` (4 aggregations, through ``SAModule``), A14 ``SAModule`` (every sampler; FRNN and ball
query), A15 ``SGCNNLayer`` (dense: 4 aggregations; sparse: max / attend, exact kNN and FRNN), A10 ``CurveSAModule`` /
``CurveFPModule``, A17 ``SharedMLP`` / ``SkipConnect``, (f)1 ``FPModule``, (f)2 ``GlobalSAModule``
(ref pointnet2.py:33-205, point_conv.py:12-93, dgcnn.py:130-266, mlp.py:5-22, skip_connect.py:6-14).
"""
from types import SimpleNamespace

import torch

from curvecloudnet_amd.synth import make_batch

NAMES = ("MLP", "SGCNNLayer", "SAModule", "CurveSAModule", "CurveFPModule", "FPModule", "GlobalSAModule", "SkipConnect",
         "SharedMLP")


def namespace(kind):
    if kind == "reference":
        from oracle.ref_import import load_reference_model
        ref = load_reference_model()
        return SimpleNamespace(MLP=ref.MLP, SGCNNLayer=ref.dgcnn.SGCNNLayer, SAModule=ref.pointnet2.SAModule,
                               CurveSAModule=ref.pointnet2.CurveSAModule, CurveFPModule=ref.pointnet2.CurveFPModule,
                               FPModule=ref.pointnet2.FPModule, GlobalSAModule=ref.pointnet2.GlobalSAModule,
                               SkipConnect=ref.skip_connect.SkipConnect, SharedMLP=ref.mlp.SharedMLP)
    if kind == "oracle":
        from oracle import torch_ref as R
        return SimpleNamespace(**{n: getattr(R, n) for n in NAMES})
    assert kind == "product"
    from curvecloudnet_amd import steps
    from curvecloudnet_amd.nn import MLP
    return SimpleNamespace(MLP=MLP, **{n: getattr(steps, n) for n in NAMES if n != "MLP"})


def _feats(n, c, seed):
    return torch.randn(n, c, generator=torch.Generator().manual_seed(seed))


def _sgcnn_dense(aggr):
    def make(ns):
        d, c = make_batch([1, 2, 3], n_curves=22), 11
        att = ns.MLP([24, 24, 24], act="leaky_relu", bias=False) if aggr in ("attend", "weighted-sum") else None
        mod = ns.SGCNNLayer(ns.MLP([2 * (c + 3), 32, 24], bias=False), 8, r=0.03, with_xyz=True, attend_nn=att, aggr_type=aggr)
        return mod, [_feats(d.pos.size(0), c, 4), d.pos, d.batch, d.curve_idxs], [0]
    return make


def _sgcnn_sparse(aggr, fast):
    def make(ns):
        d, c = make_batch([1, 2], n_curves=25), 9
        att = ns.MLP([24, 24, 24], act="leaky_relu", bias=True) if aggr == "attend" else None
        mod = ns.SGCNNLayer(ns.MLP([2 * (c + 3), 32, 24], bias=True), 12, r=0.05, with_xyz=True, attend_nn=att,
                            aggr_type=aggr, use_sparse_feat_agg=True, use_fast_knn=fast)
        return mod, [_feats(d.pos.size(0), c, 4), d.pos, d.batch, d.curve_idxs], [0]
    return make


def _sa(aggr, sampler, fast=True):
    def make(ns):
        d, c = make_batch([5, 6], n_curves=40 if sampler != "fps" else 24), 10
        att = ns.MLP([24, 12, 24], act="leaky_relu", bias=False) if aggr in ("attend", "weighted-sum") else None
        kw = dict(downsample_type=sampler, aggr_type=aggr, normalize_radius=True, attend_nn=att, use_fast_knn=fast)
        if sampler == "curve-fps":
            ratio, r, k = None, 0.05, 16
            kw["curve_fps_arclen"] = 0.012
        elif sampler == "voxel":
            ratio, r, k = None, 0.06, 16
            kw["voxel_size"] = 0.03
        elif sampler == "random":
            ratio, r, k = 0.3, 0.05, 12
        else:
            ratio, r, k = 0.25, 0.2, 16                # farthest point sampling (ball query takes K = 128 itself)
        mod = ns.SAModule(ratio, r, ns.MLP([c + 3, 32, 24], bias=False), k, **kw)
        return mod, [_feats(d.pos.size(0), c, 4), d.pos, d.batch, d.curve_idxs], [0]
    return make


def _curve_sa(curve_fps):
    def make(ns):
        d, c = make_batch([9, 10, 11] if curve_fps else [9, 10], n_curves=30), 7
        if curve_fps:
            mod = ns.CurveSAModule(None, 0.02, ns.MLP([c + 6, 24, 40], act="leaky_relu", bias=False), curve_fps_arclen=0.007,
                                   use_curve_fps=True, attend_nn=ns.MLP([40, 40, 40], act="leaky_relu", bias=False),
                                   with_xyz=True, aggr_type="attend", normalize_radius=True)
        else:
            mod = ns.CurveSAModule(0.4, 0.02, ns.MLP([c + 6, 24, 40], act="leaky_relu", bias=False), use_curve_fps=False,
                                   with_xyz=True, aggr_type="max", normalize_radius=True)
        return mod, [_feats(d.pos.size(0), c, 4), d.pos, d.batch, d.curve_idxs], [0]
    return make


def _curve_fp(ns):
    from oracle import torch_ref as R
    d, c = make_batch([9, 10, 11], n_curves=30), 7
    idx = R.curve_fps(d.pos, d.batch, d.curve_idxs, 0.007, torch.tensor([0.37]))
    mod = ns.CurveFPModule(3, ns.MLP([40 + c + 3, 32, 16], act="leaky_relu", bias=False), with_xyz=True)
    return mod, [_feats(idx.numel(), 40, 6), idx, _feats(d.pos.size(0), c, 4), d.pos, d.batch, d.curve_idxs], [0, 2]


def _fp(ns):
    d = make_batch([3, 4], n_curves=30)
    n = d.pos.size(0)
    keep = torch.arange(0, n, 3)
    mod = ns.FPModule(3, ns.MLP([12 + 5 + 3, 32, 16], bias=False), with_xyz=True)
    return mod, [_feats(keep.numel(), 12, 6), d.pos[keep], d.batch[keep], _feats(n, 5, 4), d.pos, d.batch,
                 d.curve_idxs[keep], d.curve_idxs], [0, 3]


def _global_sa(pooling):
    def make(ns):
        d, c = make_batch([5, 6, 7], n_curves=12), 6
        mod = ns.GlobalSAModule(ns.MLP([c + 3, 32, 16], bias=True), pooling=pooling)
        return mod, [_feats(d.pos.size(0), c, 4), d.pos, d.batch, d.curve_idxs], [0]
    return make


def _skip_connect(ns):
    d = make_batch([2], n_curves=40)
    n = d.pos.size(0)
    mod = ns.SkipConnect(ns.MLP([5 + 7 + 4, 24, 12], act="leaky_relu", bias=False), num_skips=2)
    return mod, [[_feats(n, 5, 1), _feats(n, 7, 2), _feats(n, 4, 3)], d.pos, d.batch, d.curve_idxs], [(0, 0), (0, 1), (0, 2)]


def _shared_mlp(ns):
    d, c = make_batch([2, 3], n_curves=30), 6
    mod = ns.SharedMLP([c + 3, 24, 16], use_bias=False, with_xyz=True, plain_last=False)
    return mod, [_feats(d.pos.size(0), c, 4), d.pos, d.batch, d.curve_idxs], [0]


CASES = {
    "sgcnn_dense_max": _sgcnn_dense("max"), "sgcnn_dense_mean": _sgcnn_dense("mean"),
    "sgcnn_dense_attend": _sgcnn_dense("attend"), "sgcnn_dense_weighted": _sgcnn_dense("weighted-sum"),
    "sgcnn_sparse_max_knn": _sgcnn_sparse("max", False), "sgcnn_sparse_attend_knn": _sgcnn_sparse("attend", False),
    "sgcnn_sparse_max_frnn": _sgcnn_sparse("max", True), "sgcnn_sparse_attend_frnn": _sgcnn_sparse("attend", True),
    "sa_curvefps_max": _sa("max", "curve-fps"), "sa_curvefps_mean": _sa("mean", "curve-fps"),
    "sa_curvefps_attend": _sa("attend", "curve-fps"), "sa_curvefps_weighted": _sa("weighted-sum", "curve-fps"),
    "sa_voxel_attend": _sa("attend", "voxel"), "sa_random_max": _sa("max", "random"),
    "sa_fps_ballquery_attend": _sa("attend", "fps", fast=False), "sa_fps_frnn_max": _sa("max", "fps", fast=True),
    "curve_sa_attend": _curve_sa(True), "curve_sa_fps_max": _curve_sa(False),
    "curve_fp": _curve_fp, "fp": _fp, "global_sa_max": _global_sa("max"), "global_sa_mean": _global_sa("mean"),
    "skip_connect": _skip_connect, "shared_mlp": _shared_mlp,
}


# The reference's dense weighted-sum reduction writes into the output of its sigmoid in place (dgcnn.py:190-192): autograd
# refuses its backward ("modified by an inplace operation"), so the reference itself can only run it forward.
FORWARD_ONLY = ("sgcnn_dense_weighted",)


def randomise_norms(module, seed=1):
    """Non-trivial BatchNorm affine parameters (a fresh BatchNorm is the identity scale: a wrong gamma path would pass)."""
    g = torch.Generator().manual_seed(seed)
    for m in module.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.weight.data.copy_(torch.rand(m.weight.shape, generator=g) + 0.5)
            m.bias.data.copy_(torch.rand(m.bias.shape, generator=g) * 0.4 - 0.2)


def cotangent(shape):
    """A fixed, generator-free cotangent (not stored in the fixtures): cos of an irrational-step ramp, O(1) entries."""
    n = 1
    for d in shape:
        n *= int(d)
    return torch.cos(torch.arange(n, dtype=torch.float64) * 0.7390851332151607 + 0.3).float().view(tuple(shape))


def _to(a, device, dtype=None):
    if isinstance(a, (list, tuple)):
        return [_to(v, device, dtype) for v in a]
    if torch.is_tensor(a):
        a = a.to(device)
        if dtype is not None and a.is_floating_point() and not (a.dim() == 2 and a.size(1) == 3):
            a = a.to(dtype)                            # (positions stay float32: index decisions are the reference's)
    return a


def run_case(module, args, diff, draws, device="cpu", dtype=None, backward=True):
    """Forward in training mode under ``draws`` (an ``oracle.draws.Draws``), cotangent from a fixed generator, gradients
    w.r.t. the ``diff`` inputs and every parameter.  Returns dict(y, outs, cot, grad_in, grad)."""
    args = _to(list(args), device, dtype)
    leaves = []
    for where in diff:
        if isinstance(where, tuple):
            args[where[0]] = list(args[where[0]])
            args[where[0]][where[1]] = args[where[0]][where[1]].clone().requires_grad_(True)
            leaves.append(args[where[0]][where[1]])
        else:
            args[where] = args[where].clone().requires_grad_(True)
            leaves.append(args[where])
    module.train()
    with draws:
        out = module(*args)
    y = out[0]
    cot = cotangent(y.shape)
    params = dict(module.named_parameters())
    if not backward:
        return dict(y=y.detach(), outs=[o.detach() if torch.is_tensor(o) else None for o in out[1:]], cot=cot, grad_in=[], grad={})
    grads = torch.autograd.grad((y * cot.to(y)).sum(), leaves + list(params.values()))
    return dict(y=y.detach(), outs=[o.detach() if torch.is_tensor(o) else None for o in out[1:]], cot=cot,
                grad_in=list(grads[: len(leaves)]), grad=dict(zip(params, grads[len(leaves):])))


# ------------------------------------------------------------------------------------------
# Whole-model cases (SURVEY.md rows A18 + H): the six shipped ``model:`` sections + the hot-path subset, reduced width
# ------------------------------------------------------------------------------------------


def _model_cases():
    from curvecloudnet_amd import configs
    return {
        # name: (config, in_dim, n_out, cloud ids, curves per cloud, which runner's loss)
        "hotpath": (lambda: configs.hotpath_config(0.25), 4, 7, [0, 1], 96, "mean"),
        "kitti": (lambda: configs.kitti_config(0.125), 4, 20, [0, 1], 150, "kitti"),
        "nuscenes": (lambda: configs.nuscenes_config(0.0625), 4, 17, [2, 3], 150, "nuscenes"),
        "a2d2": (lambda: configs.a2d2_config(0.125), 4, 13, [0, 1], 90, "a2d2"),
        "shapenet_seg": (lambda: configs.shapenet_seg_config(0.125), 3, 50, [0, 1], 90, "mean"),
        "kortx": (lambda: configs.shapenet_seg_config(0.125, kortx=True), 3, 10, [2, 3], 90, "mean"),
        "shapenet_cls": (lambda: configs.shapenet_cls_config(0.125), 3, 16, list(range(8)), 24, "mean"),
    }


MODEL_CASES = tuple(_model_cases())


def model_case(name):
    """(model kwargs, in_dim, n_out, data namespace, forward kwargs, labels, loss kind) of a whole-model case."""
    cfg, in_dim, n_out, ids, n_curves, loss = _model_cases()[name]
    kw = {k: v for k, v in cfg().items() if k != "type"}
    data = make_batch(ids, n_curves=n_curves)
    fwd = {}
    if in_dim == 3:                                   # ShapeNet / Kortx: x is None, clouds live in the unit ball
        data.x = None
        data.pos = data.pos / 3.0
    if name in ("shapenet_seg", "kortx"):
        fwd = {"shapenet-categories": torch.tensor([3, 11])}
    rows = len(ids) if name == "shapenet_cls" else data.pos.size(0)
    labels = torch.randint(0, n_out, (rows,), generator=torch.Generator().manual_seed(3))
    return kw, in_dim, n_out, data, fwd, labels, loss


# which runner's loss a case uses -> (ignore_index, reduction) of curvecloudnet_amd.model.segmentation_loss
LOSS_FORMS = {"mean": (-100, "mean"),                 # shapenet_seg.py:182-186, shapenet_classification.py:30-31
              "kitti": (0, "mean_all"),               # kitti_seg.py:184-192
              "nuscenes": (0, "mean"),                # nuscenes_seg.py:229-231 with NUSCENES_IGNORE_LABEL = 0
              "a2d2": (12, "mean")}                   # audi_seg.py:178-180 with AUDI_IGNORE_LABEL = 12


def selected_gradients(names):
    """The parameters whose full gradient a model fixture stores: the first and last three, every attend_nn's first layer,
    and every seventh in between (the others are pinned by their (sum, l2) pair)."""
    names = list(names)
    pick = set(names[:3] + names[-3:] + names[::7])
    pick.update(n for n in names if "attend_nn.lins.0" in n)
    return [n for n in names if n in pick]
