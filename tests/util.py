"""Shared helpers for the tests: golden loading, small configs, oracle <-> product weight copy."""
import copy
import os
from types import SimpleNamespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

CASES = ["one_cloud", "three_clouds", "single_points", "long_curves"]


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def t(a, device="cpu"):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def hotpath_config(width=1.0, with_sa=False):
    """Reference-style ``model:`` dict restricted to the steps of SURVEY.md section 8(a).
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


def build_pair(cfg, in_dim, n_out, seed=0):
    """(oracle model on CPU, product model) with identical weights."""
    from oracle import torch_ref as R
    from curvecloudnet_amd.model import ModelBase
    torch.manual_seed(seed)
    kw = {k: v for k, v in copy.deepcopy(cfg).items() if k != "type"}
    ref = R.ModelBase(in_dim, n_out, **copy.deepcopy(kw))
    # non-trivial BatchNorm affine parameters
    for m in ref.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.uniform_(-0.2, 0.2)
    mine = ModelBase(in_dim, n_out, **copy.deepcopy(kw))
    mine.load_state_dict(ref.state_dict(), strict=True)
    return ref, mine


def batch_to(data, device):
    out = SimpleNamespace(**vars(data))
    for k, v in vars(out).items():
        if torch.is_tensor(v):
            setattr(out, k, v.to(device))
    return out


def maxdiff(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    if a.numel() == 0:
        return 0.0
    return float((a - b).abs().max())
