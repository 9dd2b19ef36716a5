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


def _data_as(data, dtype):
    out = SimpleNamespace(**vars(data))
    if getattr(out, "x", None) is not None:
        out.x = out.x.to(dtype)
    return out


def _ref_loss(logits, labels, loss_form):
    """The runner's loss on the oracle side: ``loss_form`` = (ignore_index, "mean" | "mean_all"), oracle/module_cases.LOSS_FORMS."""
    import torch.nn.functional as F
    ignore, reduction = loss_form
    per = F.nll_loss(F.log_softmax(logits, dim=-1), labels, ignore_index=ignore, reduction="none")
    return per.mean() if reduction == "mean_all" else per.sum() / (labels != ignore).sum()


class _Seeded:
    """The random draws of one forward: torch.manual_seed(seed), or a recorded log replayed (oracle.draws.Draws)."""

    def __init__(self, seed, draws):
        self.seed, self.draws, self.ctx = seed, draws, None

    def __enter__(self):
        if self.draws is None:
            torch.manual_seed(self.seed)
        else:
            from oracle.draws import Draws
            self.ctx = Draws(replay=list(self.draws))
            self.ctx.__enter__()

    def __exit__(self, *exc):
        return self.ctx.__exit__(*exc) if self.ctx is not None else False


def routed_parity(ref, mine, data, labels, dev, seed=5, fwd_kwargs=None, backward=True, fp64=False, draws=None,
                  loss_form=(-100, "mean")):
    """Product forward (+ backward) with its routing tables recorded -- which source point won every max aggregation,
    which slope every ReLU / LeakyReLU took -- then the CPU oracle with those tables FORCED (oracle.torch_ref.MAX_TRACE,
    ACT_TRACE): both sides then differentiate along identical routes, so gradients can be held to a tight tolerance, and
    the choices the oracle would have made by itself give the number of flipped entries (all of them last-bit ties /
    pre-activations within the forward difference of zero: max_gap, sign_max_abs).

    ``fp64``: the oracle is evaluated a second time with every FEATURE computation in float64 (a ``.double()`` copy of
    the model, oracle.torch_ref.FEATURE_DTYPE; positions and every index decision stay fp32, the routes are forced to
    the same tables): ``out_64`` (and ``grad_64`` per parameter name when ``backward``) is then the value the network
    defines on this input, and the fp32 CPU oracle and the GPU are two fp32 evaluations of it whose distances to it can
    be compared (``adjudicate``).

    ``draws``: a recorded draw log (a fixture's) replayed in every forward instead of seeding torch.  ``loss_form``:
    (ignore_index, reduction) of the runner's loss on both sides.

    Returns dict(out_d, out_r, loss_d, loss_r, flips, entries, grad_err=[(err, name)], grad_scale[, out_64, grad_64])."""
    from oracle import torch_ref as R
    from curvecloudnet_amd import ops
    from curvecloudnet_amd.model import segmentation_loss
    kw = fwd_kwargs or {}
    kw_d = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in kw.items()}
    ref64 = copy.deepcopy(ref).double() if fp64 else None     # (before the passes below update the running statistics)
    ops.MAX_TRACE, ops.ACT_TRACE = [], []
    try:
        with _Seeded(seed, draws):
            out_d = mine(batch_to(data, dev), **kw_d)
    finally:
        tables, ops.MAX_TRACE = ops.MAX_TRACE, None
        signs, ops.ACT_TRACE = ops.ACT_TRACE, None
    loss_d = segmentation_loss(out_d, labels.to(dev), ignore_index=loss_form[0], reduction=loss_form[1])
    if backward:
        loss_d.backward()
    R.MAX_TRACE = {"record": [], "force": [t.clone() for t in tables]}
    R.ACT_TRACE = {"force": list(signs), "mismatch": 0, "entries": 0, "max_abs": 0.0}
    try:
        with _Seeded(seed, draws):
            out_r = ref(data, **kw)
        natural = R.MAX_TRACE["record"]
        max_gap = R.MAX_TRACE.get("max_gap", 0.0)
        act = R.ACT_TRACE
        assert not R.MAX_TRACE["force"], "the oracle ran fewer max aggregations than the product"
        assert not act["force"], "the oracle ran fewer activation layers than the product"
    finally:
        R.MAX_TRACE = R.ACT_TRACE = None
    assert len(natural) == len(tables)
    loss_r = _ref_loss(out_r, labels, loss_form)
    if backward:
        loss_r.backward()
    flips = sum(int((a != b).sum()) for a, b in zip(natural, tables))
    entries = sum(a.numel() for a in natural)
    res = dict(out_d=out_d, out_r=out_r, loss_d=loss_d, loss_r=loss_r, flips=flips, entries=entries, grad_err=[],
               grad_scale=0.0, max_gap=max_gap, sign_flips=act["mismatch"], sign_entries=act["entries"],
               sign_max_abs=act["max_abs"],
               # where the flips sit, in forward order: (kind, shape, flipped entries) per activation / |.| layer, flipped
               # entries per max aggregation
               sign_flips_per_layer=[e for e in act.get("per_layer", []) if e[2]],
               flips_per_table=[(i, tuple(a.shape), int((a != b).sum())) for i, (a, b) in enumerate(zip(natural, tables))
                                if int((a != b).sum())])
    if fp64:
        R.MAX_TRACE = {"record": [], "force": [t.clone() for t in tables]}
        R.ACT_TRACE = {"force": list(signs), "mismatch": 0, "entries": 0, "max_abs": 0.0}
        R.FEATURE_DTYPE = torch.float64
        try:
            with _Seeded(seed, draws):
                out_64 = ref64(_data_as(data, torch.float64), **kw)
            assert out_64.dtype == torch.float64
            assert not R.MAX_TRACE["force"] and not R.ACT_TRACE["force"]
            if backward:
                _ref_loss(out_64, labels, loss_form).backward()
        finally:
            R.MAX_TRACE = R.ACT_TRACE = None
            R.FEATURE_DTYPE = None
        res["out_64"] = out_64.detach()
        if backward:
            res["grad_64"] = {n: p.grad for n, p in ref64.named_parameters()}
    if backward:
        pairs = [(n, pr.grad, pd.grad) for (n, pr), (_, pd) in zip(ref.named_parameters(), mine.named_parameters())]
        for n, gr, gd in pairs:
            assert gd is not None, n
        gmax = max(float(gr.abs().max()) for _, gr, _ in pairs)
        res["grad_scale"] = gmax
        for n, gr, gd in pairs:
            # per tensor, relative to its own largest entry -- but not below 1e-3 of the model's largest gradient entry:
            # a bias in front of a BatchNorm has a mathematically zero gradient, what is left of it is summation noise
            denom = max(float(gr.abs().max()), 1e-3 * gmax)
            err = float((gd.detach().cpu() - gr).abs().max())
            if err <= 2e-6 * gmax:          # both sides at the fp32 noise floor of the summation
                err = 0.0
            res["grad_err"].append((err / denom, n))
            if fp64:
                g64 = res["grad_64"][n]
                res.setdefault("grad_adj", []).append(
                    (float((gd.detach().cpu().double() - g64).abs().max()) / denom,
                     float((gr.double() - g64).abs().max()) / denom, n))
    return res


def adjudicate(res, rms=False):
    """(d_gpu, d_cpu): max-norm (or rms) distances of the GPU logits and of the fp32 CPU oracle's logits to the fp64 evaluation."""
    o64 = res["out_64"]
    eg, ec = res["out_d"].detach().cpu().double() - o64, res["out_r"].detach().double() - o64
    if rms:
        return float(eg.square().mean().sqrt()), float(ec.square().mean().sqrt())
    return float(eg.abs().max()), float(ec.abs().max())


GRAD_TOL = 3e-4          # routed gradients, per tensor (see routed_parity)


# ---------------------------------------------------------------- round 4: fixtures generated from the reference's ModelBase
def module_fixture(name, kind, device="cpu"):
    """A case of tests/golden/modules.npz (oracle/module_cases.CASES) rebuilt from the ``kind`` side ("oracle" | "product"),
    the fixture's state_dict loaded strict, its stored inputs substituted: (module, args, diff, recorded draws, blob)."""
    from oracle import module_cases as M
    from oracle.draws import Draws
    g = golden("modules")
    mod, args, diff = M.CASES[name](M.namespace(kind))
    mod.load_state_dict({k[len(name) + 8:]: t(g[k]) for k in g.files if k.startswith(name + ".state0.")}, strict=True)
    it = iter(range(10 ** 6))
    stored = [[t(g["%s.in.%d" % (name, next(it))]) for _ in a] if isinstance(a, list) else t(g["%s.in.%d" % (name, next(it))])
              for a in args]
    return mod.to(device), stored, diff, Draws.from_blob(g, name), g


def model_fixture(name):
    """tests/golden/model_<name>.npz as (blob, model kwargs, in_dim, n_out, data, forward kwargs, labels, loss form)."""
    from oracle import module_cases as M
    g = golden("model_" + name)
    kw, in_dim, n_out, data, fwd, labels, loss_kind = M.model_case(name)
    data = SimpleNamespace(x=t(g["x"]) if "x" in g.files else None, pos=t(g["pos"]), batch=t(g["batch"]),
                           curve_idxs=t(g["curve_idxs"]), num_clouds=int(g["batch"].max()) + 1)
    fwd = {k[4:]: t(g[k]) for k in g.files if k.startswith("fwd.")}
    return g, kw, in_dim, n_out, data, fwd, t(g["labels"]), M.LOSS_FORMS[loss_kind]


def tensor_gap(got, want, floor=0.0):
    """max |got - want| relative to max(|want|_max, floor)."""
    got, want = got.detach().cpu().double(), want.detach().cpu().double()
    if want.numel() == 0:
        return 0.0
    return float((got - want).abs().max()) / max(float(want.abs().max()), floor, 1e-30)
