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


from curvecloudnet_amd.configs import hotpath_config  # noqa: E402,F401  (lives in the package: bench.py uses it)


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
