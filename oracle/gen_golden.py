"""Generates tests/golden/*.npz from the REFERENCE's own code (TEST INFRASTRUCTURE ONLY).

Run in the build container only (the reference tree is not on the GPU box):

    cd /root/repo && PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py

For every curve function on the hot path and -- round 4 -- for the step modules
and the whole ``ModelBase`` above them (rows A10, A13-A18, H: ``gen_modules`` / ``gen_models``) the reference
implementation is imported in place (``oracle/ref_import.py``), run on small seeded inputs and
its inputs/outputs are stored as arrays.  While generating, the CPU restatement in
``oracle/torch_ref.py`` is checked against the same outputs (bit-exact for integer results), so a
fixture is only written if the oracle already agrees with the reference.
The fixtures hold data only: no reference source text.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import torch_ref as R                      # noqa: E402
from oracle.ref_import import load_reference           # noqa: E402
from curvecloudnet_amd.synth import make_batch         # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def curve_case(case_id, lens_per_cloud, step=0.0035):
    """Packed batch with explicit curve lengths per cloud (covers length 1, 2, k//2, k, >k, ...)."""
    from curvecloudnet_amd.synth import make_cloud
    clouds = [make_cloud(100 * case_id + i, lengths=l, step=step) for i, l in enumerate(lens_per_cloud)]
    pos = torch.cat([c.pos for c in clouds])
    p2c = torch.cat([c.curve_idxs for c in clouds])
    batch = torch.cat([torch.full((c.pos.size(0),), i, dtype=torch.long) for i, c in enumerate(clouds)])
    x = torch.cat([c.x for c in clouds])
    return x, pos, batch, p2c


CASES = {
    # name: curve lengths per cloud
    "one_cloud": [[1, 2, 5, 7, 3, 12, 1, 1, 30, 2]],
    "three_clouds": [[4, 1, 9, 2, 2, 17, 6, 1], [1, 1, 3, 25, 8, 2, 11, 5, 1], [13, 2, 7, 1, 40, 3, 5, 9, 1]],
    "single_points": [[1, 1, 1, 1], [1, 2, 1]],
    "long_curves": [[60, 90, 33], [120, 5]],
}


def np_(t):
    return t.detach().cpu().numpy()


def eq(a, b, what):
    a, b = torch.as_tensor(a), torch.as_tensor(b)
    if a.shape != b.shape or not bool((a == b).all()):
        raise SystemExit("oracle != reference for %s" % what)


def close(a, b, what, tol=1e-5):
    err = float((torch.as_tensor(a) - torch.as_tensor(b)).abs().max()) if torch.as_tensor(a).numel() else 0.0
    if not err <= tol:
        raise SystemExit("oracle != reference for %s (max err %g)" % (what, err))


def main():
    # Bit-identical regeneration: summation order depends on the thread count (MKL / OpenMP blocking), and index_add_ --
    # the backward of every gather -- accumulates with atomics unless the deterministic algorithms are requested.
    torch.set_num_threads(8)
    torch.use_deterministic_algorithms(True)
    fc, po, fo = load_reference()
    os.makedirs(OUT, exist_ok=True)

    # ---------------- A1 / A2 / A12: index algebra ----------------
    blob = {}
    for ci, (name, lens) in enumerate(CASES.items()):
        x, pos, batch, p2c = curve_case(ci, lens)
        glob = po.curveidx_local2global(p2c.clone(), batch.clone())
        blob[name + ".batch"], blob[name + ".p2c"] = np_(batch), np_(p2c)
        blob[name + ".cloud_ptr"] = np_(po.batch2ptr(batch, with_ends=True))
        blob[name + ".cloud_ptr_interior"] = np_(po.batch2ptr(batch))
        blob[name + ".glob"] = np_(glob)
        blob[name + ".curve_ptr_interior"] = np_(po.batch2ptr(glob))
        blob[name + ".curve_ptr"] = np_(po.batch2ptr(glob, with_ends=True))
        eq(R.segment_starts(batch, True), blob[name + ".cloud_ptr"], "segment_starts")
        eq(R.curve_ids_global(p2c, batch), glob, "curve_ids_global")
        eq(R.segment_starts(glob), blob[name + ".curve_ptr_interior"], "segment_starts(glob)")
        feats = torch.cat([x, pos], dim=1)
        padded, mask, lens_b, offs = po.dense2padded_pyg(feats, batch)
        blob[name + ".feats"] = np_(feats)
        blob[name + ".padded"], blob[name + ".mask"] = np_(padded), np_(mask)
        blob[name + ".lengths"], blob[name + ".offsets"] = np_(lens_b), np_(offs)
        o = R.padded_layout(feats, batch)
        eq(o[0], padded, "padded"); eq(o[1], mask, "mask"); eq(o[2], lens_b, "lengths"); eq(o[3], offs, "offsets")
    np.savez_compressed(os.path.join(OUT, "index_algebra.npz"), **blob)

    # ---------------- A3: feature diffs (+ gradient) ----------------
    blob = {}
    for ci, (name, lens) in enumerate(CASES.items()):
        _, pos, batch, p2c = curve_case(ci, lens)
        g = torch.Generator().manual_seed(7 + ci)
        x = torch.randn(pos.size(0), 5, generator=g).requires_grad_(True)
        cot = torch.randn(pos.size(0), 5, generator=g)
        d = fc.compute_feature_diffs(x, p2c, batch)
        (gx,) = torch.autograd.grad((d * cot).sum(), x)
        blob[name + ".x"], blob[name + ".p2c"], blob[name + ".batch"] = np_(x), np_(p2c), np_(batch)
        blob[name + ".cot"], blob[name + ".diff"], blob[name + ".grad_x"] = np_(cot), np_(d), np_(gx)
        x2 = x.detach().clone().requires_grad_(True)
        d2 = R.feature_diffs(x2, p2c, batch)
        eq(d2, d, "feature_diffs")
        close(torch.autograd.grad((d2 * cot).sum(), x2)[0], gx, "feature_diffs grad", 1e-6)
    np.savez_compressed(os.path.join(OUT, "feature_diffs.npz"), **blob)

    # ---------------- A4-A6: curve convolutions ----------------
    blob = {}
    conv_cfgs = [
        ("v1_k5_diff_xyz", "v1", [4, 8, 6], 5, True, True, "three_clouds"),
        ("v1_k7_plain", "v1", [3, 6], 7, True, False, "one_cloud"),
        ("v1_k5_single", "v1", [4, 5, 5], 5, True, True, "single_points"),
        ("v2_k5_diff_xyz", "v2", [4, 8, 8, 6], 5, True, True, "three_clouds"),
        ("v2_k5_one", "v2", [4, 6, 6], 5, True, True, "one_cloud"),
        ("v2_k7_nodiff", "v2", [1, 4, 4], 7, False, False, "long_curves"),
    ]
    for tag, ver, dims, k, with_xyz, with_diff, case in conv_cfgs:
        ci = list(CASES).index(case)
        x, pos, batch, p2c = curve_case(ci, CASES[case])
        torch.manual_seed(11 + len(tag))
        cls = fc.SymmetricCurve1DConvFastV1 if ver == "v1" else fc.SymmetricCurve1DConvV2
        mine = R.SymmetricCurve1DConvFastV1 if ver == "v1" else R.SymmetricCurve1DConvV2
        ref = cls(dims, k, with_xyz=with_xyz, with_diff=with_diff)
        for nm in ref.norm_modules:
            nm.weight.data.uniform_(0.5, 1.5)
            nm.bias.data.uniform_(-0.3, 0.3)
        c_in = dims[0] - (3 if with_xyz else 0)
        feats = torch.randn(pos.size(0), c_in).requires_grad_(True) if c_in > 0 else None   # None: x = pos
        state0 = {n: v.clone() for n, v in ref.state_dict().items()}
        ref.train()
        y = ref(feats, pos, batch, p2c)[0]
        cot = torch.randn_like(y)
        params = list(ref.parameters())
        lead = [feats] if feats is not None else []
        grads = torch.autograd.grad((y * cot).sum(), lead + params)
        if feats is None:
            grads = (torch.zeros(pos.size(0), 0),) + tuple(grads)
        state1 = {n: v.clone() for n, v in ref.state_dict().items()}
        ref.eval()
        y_eval = ref(feats, pos, batch, p2c)[0]
        blob[tag + ".feats"] = np_(feats) if feats is not None else np.zeros((pos.size(0), 0), np.float32)
        blob[tag + ".pos"] = np_(pos)
        blob[tag + ".batch"], blob[tag + ".p2c"] = np_(batch), np_(p2c)
        blob[tag + ".cot"], blob[tag + ".y_train"], blob[tag + ".y_eval"] = np_(cot), np_(y), np_(y_eval)
        blob[tag + ".grad_feats"] = np_(grads[0])
        for (n, _), gval in zip(ref.named_parameters(), grads[1:]):
            blob[tag + ".grad." + n] = np_(gval)
        for n, v in state0.items():
            blob[tag + ".state0." + n] = np_(v)
        for n, v in state1.items():
            blob[tag + ".state1." + n] = np_(v)
        blob[tag + ".meta"] = np.array([int(ver[1]), k, int(with_xyz), int(with_diff)] + dims, dtype=np.int64)
        # oracle check
        m = mine(dims, k, with_xyz=with_xyz, with_diff=with_diff)
        m.load_state_dict(state0, strict=True)
        m.train()
        f2 = feats.detach().clone().requires_grad_(True) if feats is not None else None
        y2 = m(f2, pos, batch, p2c)[0]
        close(y2, y, tag + " train fwd")
        g2 = torch.autograd.grad((y2 * cot).sum(), ([f2] if f2 is not None else []) + list(m.parameters()))
        for a, b_ in zip(g2, grads if feats is not None else grads[1:]):
            close(a, b_, tag + " grads", 2e-4)
        m.eval()
        close(m(f2, pos, batch, p2c)[0], y_eval, tag + " eval fwd")
        for n, v in m.state_dict().items():
            close(v.float(), state1[n].float(), tag + " state " + n)
    np.savez_compressed(os.path.join(OUT, "curve_conv.npz"), **blob)

    # ---------------- A7: CurveFPS ----------------
    blob = {}
    for ci, (name, lens) in enumerate(CASES.items()):
        for step, spacing in ((0.0035, 0.007), (0.004, 0.03)):
            _, pos, batch, p2c = curve_case(ci, lens, step=step)
            torch.manual_seed(100 + ci)
            u = torch.rand(1)
            torch.manual_seed(100 + ci)               # the reference draws the same value itself
            idx = fo.CurveFPS(spacing)(pos.clone(), batch.clone(), p2c.clone())
            key = "%s.s%g" % (name, spacing)
            blob[key + ".pos"], blob[key + ".batch"], blob[key + ".p2c"] = np_(pos), np_(batch), np_(p2c)
            blob[key + ".u"], blob[key + ".spacing"], blob[key + ".idx"] = np_(u), np.float64(spacing), np_(idx)
            eq(R.curve_fps(pos, batch, p2c, spacing, u), idx, "curve_fps " + key)
    np.savez_compressed(os.path.join(OUT, "curve_fps.npz"), **blob)

    # ---------------- A8 / A9: curve grouping + interpolation ----------------
    blob = {}
    for ci, (name, lens) in enumerate(CASES.items()):
        _, pos, batch, p2c = curve_case(ci, lens)
        u = torch.tensor([0.37 + 0.1 * ci])
        idx = R.curve_fps(pos, batch, p2c, 0.007, u)
        for radius in (0.02, 0.006):
            row, col = po.radius_1d_group_subset(pos, idx, p2c, batch, radius)
            key = "%s.r%g" % (name, radius)
            blob[key + ".row"], blob[key + ".col"] = np_(row), np_(col)
            r2, c2 = R.curve_radius_group(pos, idx, p2c, batch, radius)
            eq(r2, row, "radius group row " + key); eq(c2, col, "radius group col " + key)
        blob[name + ".pos"], blob[name + ".batch"], blob[name + ".p2c"] = np_(pos), np_(batch), np_(p2c)
        blob[name + ".idx"] = np_(idx)
        for k in (3, 1):
            row, col = po.knn_1d_group_superset(pos, idx, p2c, batch, k)
            blob["%s.k%d.row" % (name, k)], blob["%s.k%d.col" % (name, k)] = np_(row), np_(col)
            r2, c2 = R.curve_knn_superset(pos, idx, p2c, batch, k)
            eq(r2, row, "superset row"); eq(c2, col, "superset col")
        g = torch.Generator().manual_seed(5 + ci)
        xs = torch.randn(idx.numel(), 6, generator=g).requires_grad_(True)
        cot = torch.randn(pos.size(0), 6, generator=g)
        y = po.knn_interpolate_1D_pytorch3d(xs, idx, pos, batch, p2c, 3)
        (gx,) = torch.autograd.grad((y * cot).sum(), xs)
        blob[name + ".interp_x"], blob[name + ".interp_cot"] = np_(xs), np_(cot)
        blob[name + ".interp_y"], blob[name + ".interp_grad_x"] = np_(y), np_(gx)
        xs2 = xs.detach().clone().requires_grad_(True)
        y2 = R.curve_interpolate(xs2, idx, pos, batch, p2c, 3)
        close(y2, y, "curve_interpolate", 1e-5)
        close(torch.autograd.grad((y2 * cot).sum(), xs2)[0], gx, "curve_interpolate grad", 1e-5)
    np.savez_compressed(os.path.join(OUT, "curve_group.npz"), **blob)
    # ---------------- section 8(f): VoxelFPS (reference code with the scatter_min shim) ----------------
    blob = {}
    for ci, name in enumerate(("three_clouds", "long_curves")):
        _, pos, batch, p2c = curve_case(ci + 1 if name == "three_clouds" else 3, CASES[name])
        for vs in (0.01, 0.03):
            torch.manual_seed(40 + ci)
            rnd = torch.rand(pos.size(0))
            torch.manual_seed(40 + ci)
            idx = fo.VoxelFPS(vs)(pos.clone(), batch.clone())
            key = "%s.v%g" % (name, vs)
            blob[key + ".pos"], blob[key + ".batch"], blob[key + ".rnd"] = np_(pos), np_(batch), np_(rnd)
            blob[key + ".voxel"], blob[key + ".idx"] = np.float64(vs), np_(idx)
            eq(R.voxel_fps(pos, batch, vs, rnd), idx, "voxel_fps " + key)
    np.savez_compressed(os.path.join(OUT, "voxel_fps.npz"), **blob)

    # ---------------- section 8(f) #4: dataset-side curve splitters and the Lovasz-softmax loss ----------------
    from oracle.ref_import import load_reference_harness
    lov, sem_kitti, sem_nuscenes = load_reference_harness()

    class _Self:
        CURVE_THRESH = 0.08

    blob = {}
    for ci, (n, scale, jump) in enumerate(((3000, 0.02, 0.01), (6000, 0.05, 0.03), (1, 0.0, 0.0), (2, 0.05, 0.0))):
        g = torch.Generator().manual_seed(500 + ci)
        steps = torch.randn(n, 3, generator=g) * scale
        steps[torch.rand(n, generator=g) < jump] *= 40.0          # discontinuities: new curves
        pts = (torch.cumsum(steps, 0) + torch.tensor([6.0, -3.0, 0.5])).float()
        beams = torch.randint(0, 5, (n,), generator=g)
        refl = torch.rand(n, generator=g)
        labels = torch.randint(0, 17, (n,), generator=g)
        key = "split%d" % ci
        kitti = sem_kitti._get_curves(_Self(), pts.clone()).long()
        nus = sem_nuscenes._get_curves(_Self(), pts.clone(), beams.clone(), labels.clone(), refl.clone())
        eq(R.split_curves(pts), kitti, "split_curves kitti " + key)
        mine = R.get_curves_nuscenes(pts, beams, labels, refl)
        for a, b, what in zip(mine, nus, ("points", "curves", "labels", "reflectance", "inverse")):
            eq(a, b, "get_curves_nuscenes %s %s" % (what, key))
        blob[key + ".points"], blob[key + ".beams"] = np_(pts), np_(beams)
        blob[key + ".kitti"], blob[key + ".nus_curves"], blob[key + ".nus_inverse"] = np_(kitti), np_(nus[1]), np_(nus[4])
    for ci, (n, c, absent) in enumerate(((4000, 20, 3), (1500, 7, 0), (64, 4, 2))):
        g = torch.Generator().manual_seed(600 + ci)
        probas = torch.softmax(torch.randn(n, c, generator=g) * 2.0, dim=-1).requires_grad_(True)
        labels = torch.randint(0, c - absent, (n,), generator=g)
        loss = lov.lovasz_softmax_flat(probas, labels)
        grad, = torch.autograd.grad(loss, probas)
        mine = R.lovasz_softmax_flat(probas, labels)
        close(mine, loss, "lovasz loss %d" % ci, 1e-7)
        close(torch.autograd.grad(mine, probas)[0], grad, "lovasz grad %d" % ci, 1e-7)
        key = "lovasz%d" % ci
        blob[key + ".probas"], blob[key + ".labels"] = np_(probas), np_(labels)
        blob[key + ".loss"], blob[key + ".grad"] = np_(loss), np_(grad)
    np.savez_compressed(os.path.join(OUT, "harness.npz"), **blob)
    gen_modules()
    gen_models()
    print("golden vectors written to", OUT)
    for f in sorted(os.listdir(OUT)):
        print("  %-24s %7.1f KB" % (f, os.path.getsize(os.path.join(OUT, f)) / 1024))



# ---------------------------------------------------------------------------------------------------------------
# Round 4: the assembled path from the REFERENCE's own ModelBase / step modules
# ---------------------------------------------------------------------------------------------------------------


def _state_blob(prefix, module):
    return {"%s.%s" % (prefix, n): np_(v).copy() for n, v in module.state_dict().items()}   # (copy: buffers are updated in place)


def _grad_gap(a, b):
    scale = max(float(torch.as_tensor(b).abs().max()), 1e-12)
    return float((torch.as_tensor(a) - torch.as_tensor(b)).abs().max()) / scale


def gen_modules():
    """tests/golden/modules.npz: every case of oracle/module_cases.CASES run through the reference's classes."""
    from oracle import module_cases as M
    from oracle.draws import Draws
    ns_ref, ns_or = M.namespace("reference"), M.namespace("oracle")
    blob = {}
    for name, make in M.CASES.items():
        torch.manual_seed(1000 + len(name))
        mod, args, diff = make(ns_ref)
        M.randomise_norms(mod)
        blob.update(_state_blob(name + ".state0", mod))
        rec = Draws()
        torch.manual_seed(7)
        res = M.run_case(mod, args, diff, rec, backward=name not in M.FORWARD_ONLY)
        flat = []
        for a in args:
            flat.extend(a if isinstance(a, list) else [a])
        for i, a in enumerate(flat):
            blob["%s.in.%d" % (name, i)] = np_(a)
        blob.update(rec.to_blob(name))
        blob[name + ".y"] = np_(res["y"])
        for i, o in enumerate(res["outs"]):
            if o is not None:
                blob["%s.out.%d" % (name, i + 1)] = np_(o)
        for i, gval in enumerate(res["grad_in"]):
            blob["%s.grad_in.%d" % (name, i)] = np_(gval)
        for n, gval in res["grad"].items():
            blob["%s.grad.%s" % (name, n)] = np_(gval)
        for n, v in mod.named_buffers():
            blob["%s.state1.%s" % (name, n)] = np_(v)
        # the oracle must already agree (a fixture is only written then)
        mine, args2, diff2 = make(ns_or)
        mine.load_state_dict({k[len(name) + 8:]: torch.from_numpy(v) for k, v in blob.items()
                              if k.startswith(name + ".state0.")}, strict=True)
        got = M.run_case(mine, args2, diff2, Draws(replay=rec.log), backward=name not in M.FORWARD_ONLY)
        close(got["y"], res["y"], "module %s forward" % name, 1e-5 * max(1.0, float(res["y"].abs().max())))
        for a, b in zip(got["outs"], res["outs"]):
            if b is not None:
                eq(a, b, "module %s secondary output" % name)
        for a, b in zip(got["grad_in"], res["grad_in"]):
            assert _grad_gap(a, b) < 2e-4, ("module %s input gradient" % name, _grad_gap(a, b))
        gmax = max([float(v.abs().max()) for v in res["grad"].values()] + [0.0])
        for n, b in res["grad"].items():          # (a bias in front of a BatchNorm has a mathematically zero gradient: floor)
            gap = float((got["grad"][n] - b).abs().max()) / max(float(b.abs().max()), 1e-3 * gmax, 1e-12)
            assert gap < 2e-4, ("module %s grad %s" % (name, n), gap)
        print("  module %-28s y %s  oracle |diff| %.1e" % (name, tuple(res["y"].shape), float((got["y"] - res["y"]).abs().max())))
    np.savez_compressed(os.path.join(OUT, "modules.npz"), **blob)


def reference_loss(kind, runners):
    """The loss of the reference's own runner for a model case (oracle/module_cases.LOSS_FORMS)."""
    kitti_seg, nuscenes_seg, audi_seg = runners
    if kind == "kitti":
        return lambda out, y: kitti_seg.seg_loss_kitti(out, y)[0]                               # kitti_seg.py:184-200, ignore=0
    if kind == "nuscenes":
        return lambda out, y: nuscenes_seg.seg_loss(out, y, ignore=0)[0]                        # nuscenes_seg.py:36,229-231
    if kind == "a2d2":
        return lambda out, y: audi_seg.seg_loss_audi(out, y, ignore=12)[0]                      # audi_seg.py:29,178-180
    return lambda out, y: torch.nn.functional.nll_loss(torch.log_softmax(out, dim=-1), y)      # shapenet_seg.py:182-186


def gen_models():
    """tests/golden/model_<case>.npz: the reference's ModelBase (base.py:16-215) built from the shipped model sections at
    reduced width: inputs, the random draws it took, state_dict, logits, the runner's loss, gradients, the BatchNorm
    running statistics after the step and the eval-mode logits."""
    import copy
    import yaml
    from curvecloudnet_amd import configs
    from oracle import module_cases as M
    from oracle.draws import Draws
    from oracle.ref_import import REFERENCE_ROOT, load_reference_model, load_reference_runners
    ref = load_reference_model()
    runners = load_reference_runners()
    # the programmatic sections ARE the reference's YAML sections (full width)
    base = os.path.join(REFERENCE_ROOT, "configs", "curvecloudnet-eval", "%s-curvecloudnet.yaml")
    for yml, cfg in (("kitti", configs.kitti_config()), ("nuscenes", configs.nuscenes_config()), ("audi", configs.a2d2_config()),
                     ("shapenet-seg", configs.shapenet_seg_config()), ("kortx-testsplit", configs.shapenet_seg_config(kortx=True)),
                     ("shapenet-class", configs.shapenet_cls_config())):
        assert yaml.safe_load(open(base % yml))["model"] == cfg, yml
    for name in M.MODEL_CASES:
        kw, in_dim, n_out, data, fwd, labels, loss_kind = M.model_case(name)
        torch.manual_seed(2000 + len(name))
        model = ref.base.ModelBase(in_dim, n_out, **copy.deepcopy(kw))
        M.randomise_norms(model)
        blob = _state_blob("state0", model)
        blob["pos"], blob["batch"], blob["curve_idxs"], blob["labels"] = np_(data.pos), np_(data.batch), np_(data.curve_idxs), np_(labels)
        if data.x is not None:
            blob["x"] = np_(data.x)
        for k, v in fwd.items():
            blob["fwd." + k] = np_(v)
        model.train()
        rec = Draws()
        torch.manual_seed(5)
        with rec:
            logits = model(data, **fwd)
        loss = reference_loss(loss_kind, runners)(logits, labels)
        params = dict(model.named_parameters())
        grads = dict(zip(params, torch.autograd.grad(loss, list(params.values()))))
        blob.update(rec.to_blob("train"))
        blob["logits"], blob["loss"] = np_(logits), np_(loss)
        blob["grad_names"] = np.array(list(params))
        blob["grad_summary"] = np.array([[float(g.double().sum()), float(g.double().norm())] for g in grads.values()])
        for n in M.selected_gradients(params):
            blob["grad." + n] = np_(grads[n])
        for n, v in model.named_buffers():
            blob["state1." + n] = np_(v)
        model.eval()
        rec_eval = Draws()
        with rec_eval, torch.no_grad():
            blob["logits_eval"] = np_(model(data, **fwd))
        blob.update(rec_eval.to_blob("eval"))
        # the oracle must already agree
        mine = R.ModelBase(in_dim, n_out, **copy.deepcopy(kw))
        mine.load_state_dict({k[7:]: torch.from_numpy(v) for k, v in blob.items() if k.startswith("state0.")}, strict=True)
        mine.train()
        with Draws(replay=rec.log):
            out2 = mine(data, **fwd)
        scale = max(1.0, float(logits.abs().max()))
        close(out2, logits, "model %s logits" % name, 1e-6 * scale)
        ign, red = M.LOSS_FORMS[loss_kind]
        per = torch.nn.functional.nll_loss(torch.log_softmax(out2, -1), labels, ignore_index=ign, reduction="none")
        loss2 = per.mean() if red == "mean_all" else per.sum() / (labels != ign).sum()
        close(loss2, loss, "model %s loss" % name, 1e-6)
        g2 = torch.autograd.grad(loss2, list(mine.parameters()))
        gmax = max(float(g.abs().max()) for g in grads.values())
        worst = 0.0
        for (n, b), a in zip(grads.items(), g2):
            # per tensor, relative to its own largest entry but not below 1e-3 of the model's largest (a bias in front of a
            # BatchNorm has a mathematically zero gradient: what is left is summation noise, floored at 2e-6 of the largest)
            err = float((a - b).abs().max())
            gap = 0.0 if err <= 2e-6 * gmax else err / max(float(b.abs().max()), 1e-3 * gmax)
            worst = max(worst, gap)
            assert gap <= 2e-4, ("model %s grad %s" % (name, n), gap)
        path = os.path.join(OUT, "model_%s.npz" % name)
        np.savez_compressed(path, **blob)
        print("  model %-13s %7d points, %7d parameters, %2d draws, loss %.6f, oracle logits |diff| %.1e, worst relative gradient gap %.1e"
              % (name, data.pos.size(0), sum(p.numel() for p in params.values()), len(rec.log), float(loss),
                 float((out2 - logits).abs().max()), worst))


if __name__ == "__main__":
    main()
