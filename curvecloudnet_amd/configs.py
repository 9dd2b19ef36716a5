"""Programmatic equivalents of the reference's ``model:`` YAML sections (the YAML files themselves are
not shipped).  ``kitti_config()`` reproduces configs/curvecloudnet-eval/kitti-curvecloudnet.yaml:22-426
(``nuscenes`` differs only in the first voxel size, 0.03); ``width`` scales every free channel count and
re-derives the concatenation widths of the skip-connect / fp / fp-geo steps (App. A of SURVEY.md)."""
import copy


def _w(c, width):
    return max(4, int(round(c * width)))


def _derive(steps, free, num_skips, store, width, x_channels):
    """feat_dims with the concatenation widths of skip-connect / fp / fp-geo filled in (ref base.py:159-209)."""
    names = [st if isinstance(st, str) else st["step_name"] for st in steps]
    feat_dims, w_in, prop, down = [], [x_channels], [], []     # w_in[i]: channels of x entering step i (0: x is None -> pos)
    for i, (name, dims) in enumerate(zip(names, free)):
        dims = [None if d is None else _w(d, width) for d in dims]
        cur = w_in[i]
        if name == "skip-connect":
            take = prop[-num_skips[i]:]
            del prop[-num_skips[i]:]
            dims[0] = cur + sum(w_in[j] or 3 for j in take)
        elif name in ("fp", "fp-geo"):
            j = down.pop()
            dims[0] = cur + (w_in[j] or 3) + 3
        feat_dims.append(dims)
        w_in.append(dims[-1])
        if name in store:
            prop.append(i)
        if name in ("sa", "sa-geo"):
            down.append(i)
    return feat_dims


def shapenet_seg_config(width=1.0, kortx=False):
    """configs/curvecloudnet-eval/shapenet-seg-curvecloudnet.yaml:29-363 (``kortx=True``: the
    kortx-testsplit variant: narrower front, k=7 convolutions, K=30)."""
    s = lambda **kw: dict(**kw)                                                     # noqa: E731
    sa = s(step_name="sa", aggr_type="attend", normalize_radius=True, use_fast_knn=False, downsample_type="fps")
    sg_xyz = s(step_name="sgcnn", with_xyz=True, use_fast_knn=False, use_sparse_feat_agg=True)
    sg = s(step_name="sgcnn", use_fast_knn=False, use_sparse_feat_agg=True)
    conv = s(step_name="conv1d-fast-v1", with_diff=True, with_xyz=True)
    steps = [
        s(step_name="sa-geo", curve_fps_arclen=0.04 if kortx else 0.03, use_curve_fps=True, use_curve_knn=True,
          with_xyz=True, aggr_type="attend", normalize_radius=True),
        s(step_name="mlp", plain_last=False, with_xyz=True), dict(conv), "skip-connect",
        dict(sa), dict(sg_xyz), "skip-connect", dict(sa), dict(sg_xyz), "skip-connect", dict(sa), dict(sg_xyz), dict(sg),
        "skip-connect",
        s(step_name="fp", with_xyz=True), dict(sg), "skip-connect",
        s(step_name="fp", with_xyz=True), dict(sg), "skip-connect",
        s(step_name="fp", with_xyz=True), dict(conv), "skip-connect",
        s(step_name="fp-geo", with_xyz=True), dict(conv), "skip-connect",
    ]
    if kortx:
        free = [[64, 128, 256, 512], [256, 128, 64], [64, 64], [None, 128], [128, 128, 128], [128, 128], [None, 256],
                [256, 256, 256], [256, 256], [None, 512], [512, 512, 512], [512, 512], [512, 512], [None, 1024, 512],
                [None, 512, 256], [256, 256], [None, 512, 256], [None, 256, 128], [128, 128], [None, 256, 128],
                [None, 128, 64], [64, 48], [None, 64, 64], [None, 64, 64], [64, 64], [None, 64, 64]]
    else:
        free = [[64, 128, 256, 512, 1024], [512, 256, 128], [128, 128], [None, 128], [128, 128, 128], [128, 128],
                [None, 256], [256, 256, 256], [256, 256], [None, 512], [512, 512, 512], [512, 512], [512, 512],
                [None, 1024, 512], [None, 512, 256], [256, 256], [None, 512, 256], [None, 256, 128], [128, 128],
                [None, 256, 128], [None, 128, 128], [128, 128], [None, 128, 128], [None, 128, 128], [64, 64],
                [None, 128, 64]]
    num_skips = [None, None, None, 1, None, None, 1, None, None, 1, None, None, None, 2, None, None, 1, None, None, 1,
                 None, None, 1, None, None, 1]
    store = ["conv1d-fast-v1", "sgcnn"]
    k = 30 if kortx else 20
    ks = 7 if kortx else 5
    return dict(
        type="generic", use_bias=True, version=1.0, steps=copy.deepcopy(steps),
        feat_dims=_derive(steps, free, num_skips, store, width, 0),
        out_mlp={"dims": [_w(64, width), _w(64, width)], "dropout": 0.0, "with_seg_category": True},
        knn=[None, None, k, None, None, k, None, None, k, None, None, k, k, k, 3, k, None, 3, k, None, 3, k, None, 3, None,
             None],
        ratios=[None] * 4 + [0.25, None, None, 0.25, None, None, 0.25 if kortx else 0.5] + [None] * 15,
        radii=([0.075, None, None, None, 0.2, None, None, 0.4, None, None, 0.8] if kortx else
               [0.04, None, None, None, 0.18, None, None, 0.35, None, None, 0.7]) + [None] * 15,
        num_skips=num_skips,
        kernel_sizes=[None, None, ks] + [None] * 18 + [ks, None, None, ks, None],
        skip_connect_state_store=store,
    )


def a2d2_config(width=1.0):
    """configs/curvecloudnet-eval/audi-curvecloudnet.yaml (in_dim 4; FRNN + attention in the sparse SGCNN steps)."""
    s = lambda **kw: dict(**kw)                                                     # noqa: E731
    sa = s(step_name="sa", aggr_type="attend", normalize_radius=True, use_fast_knn=False, downsample_type="fps")
    sg_xyz = s(step_name="sgcnn", with_xyz=True, aggr_type="attend", use_sparse_feat_agg=True)
    sg = s(step_name="sgcnn", aggr_type="attend", use_sparse_feat_agg=True)
    conv = s(step_name="conv1d-fast-v1", with_diff=True, with_xyz=True)
    steps = [
        s(step_name="sa-geo", curve_fps_arclen=0.01, use_curve_fps=True, use_curve_knn=True, with_xyz=True,
          aggr_type="attend", normalize_radius=True),
        s(step_name="mlp", plain_last=False, with_xyz=True), dict(conv), "skip-connect",
        dict(sa), dict(sg_xyz), "skip-connect", dict(sa), dict(sg_xyz), "skip-connect", dict(sa), dict(sg_xyz), dict(sg),
        "skip-connect",
        s(step_name="fp", with_xyz=True), dict(sg), "skip-connect",
        s(step_name="fp", with_xyz=True), dict(sg), "skip-connect",
        s(step_name="fp", with_xyz=True), dict(conv), "skip-connect",
        s(step_name="fp-geo", with_xyz=True), dict(conv), "skip-connect",
    ]
    free = [[64, 128, 256, 512], [256, 128, 64], [64, 64], [None, 128], [128, 128, 128], [128, 128], [None, 256],
            [256, 256, 256], [256, 256], [None, 512], [512, 512, 512], [512, 512], [512, 512], [None, 1024, 512],
            [None, 512, 256], [256, 256], [None, 512, 256], [None, 256, 128], [128, 128], [None, 256, 128],
            [None, 128, 64], [64, 64], [None, 64, 64], [None, 128, 128], [128, 128], [None, 128, 64]]
    num_skips = [None, None, None, 1, None, None, 1, None, None, 1, None, None, None, 2, None, None, 1, None, None, 1,
                 None, None, 1, None, None, 1]
    store = ["conv1d-fast-v1", "sgcnn"]
    return dict(
        type="generic", use_bias=True, version=1.0, steps=copy.deepcopy(steps),
        feat_dims=_derive(steps, free, num_skips, store, width, 1),
        out_mlp={"dims": [_w(64, width), _w(64, width)], "dropout": 0.0},
        knn=[None, None, 30, None, None, 30, None, None, 30, None, None, 30, 30, 30, 3, 30, None, 3, 30, None, 3, 30, None,
             3, None, None],
        ratios=[None] * 4 + [0.35, None, None, 0.25, None, None, 0.25] + [None] * 15,
        radii=[0.015, None, 0.1, None, 0.03, 0.25, None, 0.06, 0.5, None, 0.15, 1.5, 1.5, None, None, 0.5, None, None, 0.25,
               None, None, 0.1, None, None, None, None],
        num_skips=num_skips,
        kernel_sizes=[None, None, 5] + [None] * 18 + [5, None, None, 5, None],
        skip_connect_state_store=store,
    )


def shapenet_cls_config(width=1.0):
    """configs/curvecloudnet-eval/shapenet-class-curvecloudnet.yaml (15 steps, ends in the global pooling step)."""
    full = shapenet_seg_config(1.0, kortx=True)
    steps = copy.deepcopy(full["steps"][:14]) + ["sa-global"]
    free = [[64, 128, 256, 512], [256, 128, 64], [64, 64], [None, 128], [128, 128, 128], [128, 128], [None, 256],
            [256, 256, 256], [256, 256], [None, 512], [512, 512, 512], [512, 512], [512, 512], [None, 1024, 1024],
            [1024, 1024]]
    num_skips = [None, None, None, 1, None, None, 1, None, None, 1, None, None, None, 2, None]
    store = ["conv1d-fast-v1", "sgcnn"]
    return dict(
        type="generic", use_bias=True, version=1.0, steps=steps,
        feat_dims=_derive(steps, free, num_skips, store, width, 0),
        out_mlp={"dims": [_w(512, width), _w(256, width), _w(128, width)], "dropout": 0.0, "with_seg_category": False},
        knn=[None, None, 30, None, None, 30, None, None, 30, None, None, 30, 30, 30, None],
        ratios=[None] * 4 + [0.25, None, None, 0.25, None, None, 0.25] + [None] * 4,
        radii=[0.075, None, None, None, 0.2, None, None, 0.4, None, None, 0.8] + [None] * 4,
        num_skips=num_skips,
        kernel_sizes=[None, None, 7] + [None] * 12,
        skip_connect_state_store=store,
    )


def kitti_config(width=1.0, first_voxel=0.025, in_dim=4):
    s = lambda **kw: dict(**kw)                                                     # noqa: E731
    sa_vox = lambda v: s(step_name="sa", aggr_type="attend", downsample_type="voxel", voxel_size=v,   # noqa: E731
                         normalize_radius=True, use_fast_knn=True)
    sa_fps = lambda a: s(step_name="sa", aggr_type=a, downsample_type="fps", normalize_radius=True,   # noqa: E731
                         use_fast_knn=True)
    sg_xyz = s(step_name="sgcnn", with_xyz=True, aggr_type="max")
    steps = [
        s(step_name="conv1d-fast-v2", with_diff=True, with_xyz=True),
        s(step_name="sa-geo", curve_fps_arclen=0.007, use_curve_fps=True, use_curve_knn=True, with_xyz=True,
          aggr_type="attend", normalize_radius=True),
        s(step_name="mlp", plain_last=False, with_xyz=True),
        dict(sg_xyz), "skip-connect", sa_vox(first_voxel),
        dict(sg_xyz), "skip-connect", sa_vox(0.07),
        dict(sg_xyz), "skip-connect", sa_fps("attend"),
        dict(sg_xyz), "skip-connect", sa_fps("max"),
        dict(sg_xyz), s(step_name="sgcnn", aggr_type="max"), "skip-connect",
        s(step_name="fp", with_xyz=True), "sgcnn", "skip-connect",
        s(step_name="fp", with_xyz=True), "sgcnn", "skip-connect",
        s(step_name="fp", with_xyz=True), "sgcnn", "skip-connect",
        s(step_name="fp", with_xyz=True), dict(sg_xyz), "skip-connect",
        s(step_name="fp-geo", with_xyz=True),
        s(step_name="conv1d-fast-v2", with_diff=True, with_xyz=True),
        "skip-connect",
    ]
    # free (non-derived) channel counts of every step; None marks a width derived from the concatenation
    free = [
        [32, 32, 32], [64, 128, 192, 256], [256, 128, 128, 64],
        [64, 64, 64], [None, 128, 128], [128, 128, 128],
        [128, 128], [None, 256], [256, 256, 256],
        [256, 256], [None, 512], [512, 512, 512],
        [512, 512], [None, 1024], [1024, 1024, 1024],
        [1024, 1024], [1024, 1024], [None, 2048, 1024],
        [None, 1024, 512], [512, 512], [None, 1024, 512],
        [None, 512, 256], [256, 256], [None, 512, 256],
        [None, 256, 128], [128, 128], [None, 256, 128],
        [None, 128, 64], [64, 64, 64], [None, 64, 64],
        [None, 128, 128], [32, 32, 32], [None, 128, 64],
    ]
    num_skips = [None, None, None, None, 1, None, None, 1, None, None, 1, None, None, 1, None, None, None, 2, None,
                 None, 1, None, None, 1, None, None, 1, None, None, 1, None, None, 1]
    store = ["conv1d-fast-v2", "sgcnn"]
    names = [st if isinstance(st, str) else st["step_name"] for st in steps]
    feat_dims, w_in, prop, down = [], [in_dim - 3], [], []      # w_in[i]: channels of x entering step i
    for i, (name, dims) in enumerate(zip(names, free)):
        dims = [None if d is None else _w(d, width) for d in dims]
        cur = w_in[i]
        if name == "skip-connect":
            take = prop[-num_skips[i]:]
            del prop[-num_skips[i]:]
            dims[0] = cur + sum(w_in[j] for j in take)
        elif name in ("fp", "fp-geo"):
            j = down.pop()
            dims[0] = cur + w_in[j] + 3
        feat_dims.append(dims)
        w_in.append(dims[-1])
        if name in store:
            prop.append(i)
        if name in ("sa", "sa-geo"):
            down.append(i)
    return dict(
        type="generic", use_bias=False, version=2.0, steps=copy.deepcopy(steps), feat_dims=feat_dims,
        out_mlp={"dims": [_w(64, width), _w(64, width)], "dropout": 0.0},
        knn=[None, None, None, 20, None, 32, 20, None, 32, 20, None, 32, 20, None, 32, 20, 20, 20, 3, 20, None, 3, 20,
             None, 3, 20, None, 3, 20, None, 3, 8, None],
        ratios=[None] * 5 + [0.3, None, None, 0.3, None, None, 0.3, None, None, 0.3] + [None] * 18,
        radii=[None, 0.02, None, 0.04, None, 0.04, 0.08, None, 0.1, 0.3, None, 0.3, 0.3, None, 0.5, 0.8, 0.8, None, None,
               0.3, None, None, 0.3, None, None, 0.08, None, None, 0.04, None, None, 0.02, None],
        num_skips=num_skips,
        kernel_sizes=[5, None, None, 3] + [None] * 24 + [3, None, None, 5, None],
        skip_connect_state_store=store,
    )


def nuscenes_config(width=1.0):
    return kitti_config(width, first_voxel=0.03)


def hotpath_config(width=1.0, with_sa=False):
    """Reference-style ``model:`` dict restricted to the steps of SURVEY.md section 8(a) (no voxel / FPS levels,
    no exact-kNN up-sampling): the KITTI front end conv1d-fast-v2 -> sa-geo -> mlp -> 2 x (sgcnn, skip) -> fp-geo -> conv -> skip.
    width=1.0 gives the KITTI/nuScenes channel counts of App. A for those steps."""
    def w(c):
        return max(4, int(round(c * width)))
    steps = [
        {"step_name": "conv1d-fast-v2", "with_diff": True, "with_xyz": True},
        {"step_name": "sa-geo", "curve_fps_arclen": 0.007, "use_curve_fps": True, "use_curve_knn": True,
         "with_xyz": True, "aggr_type": "attend", "normalize_radius": True},
        {"step_name": "mlp", "plain_last": False, "with_xyz": True},
        {"step_name": "sgcnn", "with_xyz": True, "aggr_type": "max"},
        "skip-connect",
        {"step_name": "sgcnn", "with_xyz": True, "aggr_type": "max"},
        "skip-connect",
        {"step_name": "fp-geo", "with_xyz": True},
        {"step_name": "conv1d-fast-v2", "with_diff": True, "with_xyz": True},
        "skip-connect",
    ]
    feat_dims = [
        [w(32), w(32), w(32)],
        [w(64), w(128), w(192), w(256)],
        [w(256), w(128), w(128), w(64)],
        [w(64), w(64), w(64)],
        [2 * w(64), w(128), w(128)],
        [w(128), w(128)],
        [2 * w(128), w(128), w(64)],
        [w(64) + w(32) + 3, w(128), w(128)],
        [w(32), w(32), w(32)],
        [w(32) + w(128), w(128), w(64)],
    ]
    n = len(steps)
    cfg = dict(
        type="generic", use_bias=False, version=2.0, steps=steps, feat_dims=feat_dims,
        out_mlp={"dims": [w(64), w(64)], "dropout": 0.0},
        knn=[None, None, None, 20, None, 20, None, 3, None, None],
        ratios=[None] * n,
        radii=[None, 0.02, None, 0.04, None, 0.08, None, None, None, None],
        num_skips=[None, None, None, None, 1, None, 1, None, None, 1],
        kernel_sizes=[5, None, None, None, None, None, None, None, 5, None],
        skip_connect_state_store=["conv1d-fast-v2", "sgcnn"],
    )
    return cfg
