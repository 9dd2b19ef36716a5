"""CPU: the oracle restatement against the golden vectors generated from the reference's own code
(oracle/gen_golden.py).  Integer results bit-exact, floats to 1e-5."""
import numpy as np
import pytest
import torch

from oracle import torch_ref as R
from tests.util import CASES, golden, t


@pytest.mark.parametrize("case", CASES)
def test_index_algebra(case):
    g = golden("index_algebra")
    batch, p2c = t(g[case + ".batch"]), t(g[case + ".p2c"])
    assert torch.equal(R.segment_starts(batch, True), t(g[case + ".cloud_ptr"]))
    assert torch.equal(R.segment_starts(batch), t(g[case + ".cloud_ptr_interior"]))
    glob = R.curve_ids_global(p2c, batch)
    assert torch.equal(glob, t(g[case + ".glob"]))
    assert torch.equal(R.segment_starts(glob, True), t(g[case + ".curve_ptr"]))
    padded, mask, lens, offs = R.padded_layout(t(g[case + ".feats"]), batch)
    assert torch.equal(padded, t(g[case + ".padded"])) and torch.equal(mask, t(g[case + ".mask"]))
    assert torch.equal(lens, t(g[case + ".lengths"])) and torch.equal(offs, t(g[case + ".offsets"]))


def test_single_cloud_returns_input_object():
    p2c = torch.tensor([0, 0, 1, 2])
    assert R.curve_ids_global(p2c, torch.zeros(4, dtype=torch.long)) is p2c          # quirk Q8
    with pytest.raises(AssertionError):
        R.segment_starts(torch.tensor([1, 0]))


@pytest.mark.parametrize("case", CASES)
def test_feature_diffs(case):
    g = golden("feature_diffs")
    x = t(g[case + ".x"]).requires_grad_(True)
    d = R.feature_diffs(x, t(g[case + ".p2c"]), t(g[case + ".batch"]))
    assert torch.equal(d.detach(), t(g[case + ".diff"]))
    (gx,) = torch.autograd.grad((d * t(g[case + ".cot"])).sum(), x)
    assert float((gx - t(g[case + ".grad_x"])).abs().max()) < 1e-6


def test_curve_conv_all_cases():
    g = golden("curve_conv")
    tags = sorted({k.split(".")[0] for k in g.files})
    assert len(tags) == 6
    for tag in tags:
        meta = g[tag + ".meta"].tolist()
        ver, k, with_xyz, with_diff, dims = meta[0], meta[1], bool(meta[2]), bool(meta[3]), meta[4:]
        cls = R.SymmetricCurve1DConvFastV1 if ver == 1 else R.SymmetricCurve1DConvV2
        m = cls(dims, k, with_xyz=with_xyz, with_diff=with_diff)
        m.load_state_dict({n[len(tag) + 8:]: t(g[n]) for n in g.files if n.startswith(tag + ".state0.")}, strict=True)
        m.train()
        feats = t(g[tag + ".feats"]).requires_grad_(True) if g[tag + ".feats"].shape[1] else None   # None: x = pos
        args = (t(g[tag + ".pos"]), t(g[tag + ".batch"]), t(g[tag + ".p2c"]))
        y = m(feats, *args)[0]
        assert float((y - t(g[tag + ".y_train"])).abs().max()) < 1e-5, tag
        lead = [feats] if feats is not None else []
        grads = torch.autograd.grad((y * t(g[tag + ".cot"])).sum(), lead + list(m.parameters()))
        if feats is not None:
            assert float((grads[0] - t(g[tag + ".grad_feats"])).abs().max()) < 2e-4, tag
        for (n, _), gv in zip(m.named_parameters(), grads[len(lead):]):
            assert float((gv - t(g[tag + ".grad." + n])).abs().max()) < 2e-4, (tag, n)
        m.eval()
        assert float((m(feats, *args)[0] - t(g[tag + ".y_eval"])).abs().max()) < 1e-5, tag


def test_curve_fps():
    g = golden("curve_fps")
    keys = sorted({k.rsplit(".", 1)[0] for k in g.files})
    assert len(keys) == 8
    for key in keys:
        idx = R.curve_fps(t(g[key + ".pos"]), t(g[key + ".batch"]), t(g[key + ".p2c"]), float(g[key + ".spacing"]),
                          t(g[key + ".u"]))
        assert torch.equal(idx, t(g[key + ".idx"])), key


@pytest.mark.parametrize("case", CASES)
def test_curve_groups_and_interpolation(case):
    g = golden("curve_group")
    pos, batch, p2c, idx = (t(g[case + s]) for s in (".pos", ".batch", ".p2c", ".idx"))
    for radius in (0.02, 0.006):
        row, col = R.curve_radius_group(pos, idx, p2c, batch, radius)
        key = "%s.r%g" % (case, radius)
        assert torch.equal(row, t(g[key + ".row"])) and torch.equal(col, t(g[key + ".col"]))
    for k in (3, 1):
        row, col = R.curve_knn_superset(pos, idx, p2c, batch, k)
        assert torch.equal(row, t(g["%s.k%d.row" % (case, k)])) and torch.equal(col, t(g["%s.k%d.col" % (case, k)]))
    x = t(g[case + ".interp_x"]).requires_grad_(True)
    y = R.curve_interpolate(x, idx, pos, batch, p2c, 3)
    assert float((y - t(g[case + ".interp_y"])).abs().max()) < 1e-5
    (gx,) = torch.autograd.grad((y * t(g[case + ".interp_cot"])).sum(), x)
    assert float((gx - t(g[case + ".interp_grad_x"])).abs().max()) < 1e-5


def test_voxel_fps():
    g = golden("voxel_fps")
    keys = sorted({k.rsplit(".", 1)[0] for k in g.files})
    assert len(keys) == 4
    for key in keys:
        idx = R.voxel_fps(t(g[key + ".pos"]), t(g[key + ".batch"]), float(g[key + ".voxel"]), t(g[key + ".rnd"]))
        assert torch.equal(idx, t(g[key + ".idx"])), key


def test_farthest_points_and_exact_knn_known_answers():
    # a line of points: FPS from index 0 picks the far end, then the middle
    pos = torch.stack([torch.arange(9, dtype=torch.float32), torch.zeros(9), torch.zeros(9)], 1)
    idx = R.farthest_point_indices(pos, torch.zeros(9, dtype=torch.long), 3 / 9, start=[0])
    assert idx.tolist() == [0, 4, 8]
    nn = R.knn_bruteforce(pos[None], pos[None], torch.tensor([9]), torch.tensor([9]), 3)
    assert nn[0, 0].tolist() == [0, 1, 2] and nn[0, 4].tolist() == [4, 3, 5]      # ties: smaller index first
    y = R.knn_interpolate(pos[:, :1].clone(), pos, pos + 0.25, torch.zeros(9, dtype=torch.long),
                          torch.zeros(9, dtype=torch.long), 1)
    assert torch.allclose(y, pos[:, :1])


def test_frnn_bruteforce_known_answers():
    """FRNN is third party and un-vendored (parity unpinned): anchor the exhaustive oracle on
    hand-checkable cases."""
    ax = torch.arange(4, dtype=torch.float32)
    lat = torch.stack(torch.meshgrid(ax, ax, ax, indexing="ij"), -1).reshape(1, -1, 3)
    n = torch.tensor([64])
    idx, d2 = R.frnn_bruteforce(lat, lat, n, n, 8, 1.1, return_dists=True)
    centre = 21                                   # (1,1,1): itself + 6 axis neighbours at distance 1
    assert idx[0, centre].tolist() == [21, 5, 17, 20, 22, 25, 37, -1]
    assert d2[0, centre].tolist() == [0, 1, 1, 1, 1, 1, 1, -1]
    assert idx[0, 0].tolist() == [0, 1, 4, 16, -1, -1, -1, -1]        # corner
    idx = R.frnn_bruteforce(lat, lat, n, n, 3, 1.0)                   # strict d2 < r*r: only itself
    assert (idx[0, :, 0] == torch.arange(64)).all() and (idx[0, :, 1:] == -1).all()
    idx = R.frnn_bruteforce(lat, lat, torch.tensor([2]), n, 2, 5.0)   # rows >= lengths1 are -1
    assert (idx[0, 2:] == -1).all() and idx[0, 1].tolist() == [1, 0]
    # against a dense numpy computation on random points
    gen = torch.Generator().manual_seed(0)
    p = torch.rand(2, 120, 3, generator=gen)
    l = torch.tensor([120, 77])
    got = R.frnn_bruteforce(p, p, l, l, 6, 0.25)
    for b in range(2):
        pts = p[b, : l[b]].numpy().astype(np.float64)
        dm = ((pts[:, None] - pts[None]) ** 2).sum(-1)
        for i in range(int(l[b])):
            order = [j for j in np.argsort(dm[i], kind="stable") if dm[i, j] < 0.25 ** 2 * (1 - 1e-6)][:6]
            assert got[b, i, : len(order)].tolist() == order


def test_curve_splitters_against_reference_vectors():
    """Oracle restatement of the dataset-side splitters vs. the reference's own outputs (harness.npz)."""
    g = golden("harness")
    for ci in range(4):
        key = "split%d" % ci
        pts, beams = t(g[key + ".points"]), t(g[key + ".beams"])
        assert torch.equal(R.split_curves(pts), t(g[key + ".kitti"]))
        out = R.get_curves_nuscenes(pts, beams, torch.zeros(len(pts)), torch.zeros(len(pts)))
        assert torch.equal(out[1], t(g[key + ".nus_curves"]))
        assert torch.equal(out[4], t(g[key + ".nus_inverse"]))


@pytest.mark.parametrize("impl", ["oracle", "product"])
def test_lovasz_softmax_against_reference_vectors(impl):
    """Both the oracle's per-class loop and the harness' all-classes-at-once form (curvecloudnet_amd.loss, plain
    torch ops, device agnostic) reproduce the reference's loss and gradient."""
    from curvecloudnet_amd import loss as L
    fn = R.lovasz_softmax_flat if impl == "oracle" else L.lovasz_softmax_flat
    g = golden("harness")
    for ci in range(3):
        key = "lovasz%d" % ci
        probas = t(g[key + ".probas"]).requires_grad_(True)
        out = fn(probas, t(g[key + ".labels"]))
        grad, = torch.autograd.grad(out, probas)
        assert abs(float(out) - float(g[key + ".loss"])) < 1e-6
        assert float((grad - t(g[key + ".grad"])).abs().max()) < 1e-7
    pred = torch.randn(700, 20, generator=torch.Generator().manual_seed(3), requires_grad=True)
    gt = torch.randint(0, 20, (700,), generator=torch.Generator().manual_seed(4))
    for kw in (dict(), dict(use_lovasz=True), dict(use_lovasz=True, class_weights=torch.linspace(0.5, 2.0, 19))):
        a, pa = L.seg_loss_kitti(pred, gt, **kw)
        b, pb = R.seg_loss_kitti(pred, gt, **kw)
        assert abs(float(a) - float(b)) < 1e-6 and torch.allclose(pa, pb, atol=1e-6)


# ---------------------------------------------------------------- round 4: the reference's own step modules and ModelBase
def _module_names():
    from oracle import module_cases as M
    return list(M.CASES)


@pytest.mark.parametrize("name", _module_names())
def test_step_modules_against_reference_vectors(name):
    """tests/golden/modules.npz was produced by the REFERENCE's classes (pointnet2.py, point_conv.py, dgcnn.py, mlp.py,
    skip_connect.py imported in place, oracle/ref_import.py); the oracle restatement reproduces output, sampled indices /
    positions, input and parameter gradients and the BatchNorm running statistics."""
    from oracle import module_cases as M
    from oracle.draws import Draws
    from tests.util import module_fixture, tensor_gap
    mod, args, diff, draws, g = module_fixture(name, "oracle")
    res = M.run_case(mod, args, diff, Draws(replay=draws), backward=name not in M.FORWARD_ONLY)
    want = t(g[name + ".y"])
    assert res["y"].shape == want.shape and tensor_gap(res["y"], want, 1.0) <= 1e-6, name
    for i, o in enumerate(res["outs"]):
        key = "%s.out.%d" % (name, i + 1)
        assert (o is None) == (key not in g.files), key
        if o is not None:
            assert torch.equal(o, t(g[key])), key               # positions, batch, curve ids, sampled indices: exact
    gmax = max([float(np.abs(g[k]).max()) for k in g.files if k.startswith(name + ".grad.")] + [0.0])
    for i, gi in enumerate(res["grad_in"]):
        assert tensor_gap(gi, t(g["%s.grad_in.%d" % (name, i)]), 1e-3 * gmax) <= 1e-4, (name, i)
    for n, gp in res["grad"].items():
        assert tensor_gap(gp, t(g["%s.grad.%s" % (name, n)]), 1e-3 * gmax) <= 1e-4, (name, n)
    for n, b in mod.named_buffers():
        assert tensor_gap(b.float(), t(g["%s.state1.%s" % (name, n)]).float(), 1.0) <= 1e-6, (name, n)


def _model_names():
    from oracle import module_cases as M
    return list(M.MODEL_CASES)


@pytest.mark.parametrize("name", _model_names())
def test_model_sections_against_reference_vectors(name):
    """tests/golden/model_<name>.npz was produced by the REFERENCE's ``ModelBase`` (base.py:16-215 imported in place) on
    the shipped model sections at reduced width, with the loss of the reference's own runner.  The oracle
    with the fixture's state_dict (strict) and the recorded draws reproduces logits, loss, every parameter gradient
    (selected tensors in full, all of them by their sum and l2 norm), running statistics and the eval-mode logits."""
    import copy
    import torch.nn.functional as F
    from oracle import module_cases as M
    from oracle.draws import Draws
    from tests.util import model_fixture, tensor_gap
    g, kw, in_dim, n_out, data, fwd, labels, (ignore, reduction) = model_fixture(name)
    model = R.ModelBase(in_dim, n_out, **copy.deepcopy(kw))
    model.load_state_dict({k[7:]: t(g[k]) for k in g.files if k.startswith("state0.")}, strict=True)
    model.train()
    with Draws(replay=Draws.from_blob(g, "train")):
        logits = model(data, **fwd)
    want = t(g["logits"])
    assert logits.shape == want.shape and tensor_gap(logits, want, 1.0) <= 1e-6
    per = F.nll_loss(F.log_softmax(logits, -1), labels, ignore_index=ignore, reduction="none")
    loss = per.mean() if reduction == "mean_all" else per.sum() / (labels != ignore).sum()
    assert abs(float(loss) - float(g["loss"])) <= 1e-6
    names = [str(n) for n in g["grad_names"]]
    assert names == [n for n, _ in model.named_parameters()]
    grads = dict(zip(names, torch.autograd.grad(loss, list(model.parameters()))))
    gmax = float(np.abs(g["grad_summary"][:, 1]).max())
    stored = [k[5:] for k in g.files if k.startswith("grad.")]
    assert len(stored) >= 10 and stored == M.selected_gradients(names)
    for n in stored:
        assert tensor_gap(grads[n], t(g["grad." + n]), 1e-3 * gmax) <= 2e-4, n
    for (n, gr), (s, l2) in zip(grads.items(), g["grad_summary"]):
        assert abs(float(gr.double().norm()) - l2) <= 2e-4 * max(l2, 1e-3 * gmax), n
        assert abs(float(gr.double().sum()) - s) <= 2e-4 * max(l2, 1e-3 * gmax) * max(1.0, gr.numel() ** 0.5), n
    for n, b in model.named_buffers():
        assert tensor_gap(b.float(), t(g["state1." + n]).float(), 1.0) <= 1e-6, n
    model.eval()
    with Draws(replay=Draws.from_blob(g, "eval")), torch.no_grad():
        assert tensor_gap(model(data, **fwd), t(g["logits_eval"]), 1.0) <= 1e-6
