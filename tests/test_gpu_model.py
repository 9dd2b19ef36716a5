"""GPU parity of the assembled hot path (ModelBase over the section-8a steps) against the CPU oracle:
logits, loss, gradients of every parameter, BatchNorm running statistics -- plus properties at the
BASELINE size (2048 curves, ~50k points)."""
import os

import pytest
import torch

from tests.util import GRAD_TOL, adjudicate, batch_to, build_pair, hotpath_config, maxdiff, routed_parity

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _labels(n, classes, seed):
    return torch.randint(0, classes, (n,), generator=torch.Generator().manual_seed(seed))


LOGIT_TOL = 1e-4        # north_star: fp32 features within 1e-4 (times the logit scale where that exceeds 1)
MAX_FLIP_RATE = 1e-4    # arg-max entries that may differ between GPU and oracle (last-bit ties), plus 4


ADJ_RATIO = 1.5         # adjudicated by fp64: the GPU may be at most this much farther from the fp64 value than the fp32 CPU oracle


RUN_LABEL = ""          # prefix of every margins line: "[bf16x3] " while tests/test_gpu_gemm_x3.py re-runs these tests in that mode


def _log(line):
    """Printed (pytest -s) and, when CCN_PARITY_LOG names a file, appended there (profiles/rNN_parity_margins.txt)."""
    line = RUN_LABEL + line
    print(line)
    path = os.environ.get("CCN_PARITY_LOG")
    if path:
        with open(path, "a") as f:
            f.write(line + "\n")


def _check_routed(res, what):
    """Shared assertions of a tests.util.routed_parity run (see there).

    Logits: within north_star's 1e-4 (times the logit scale where that exceeds 1) of the fp32 CPU oracle -- or, when the
    run carries an fp64 evaluation of the oracle along the same routes (``fp64=True``), ADJUDICATED by it: two correct
    fp32 evaluations of a deep network differ from each other by the sum of their rounding errors, so the GPU is held
    to  |gpu - fp64| <= max(1e-4 scale, 1.5 |cpu_fp32 - fp64|)  and  |gpu - fp64| <= 1e-4 scale + |cpu_fp32 - fp64|  (max norm):
    as close to the value the network defines as the reference's own arithmetic is.  Gradients likewise: every tensor
    within GRAD_TOL of the fp32 oracle, or no farther from the fp64 gradient than 1.5 x the fp32 oracle is (+ GRAD_TOL/3)."""
    out_d, out_r = res["out_d"], res["out_r"]
    assert out_d.shape == out_r.shape
    scale = max(1.0, float(out_r.abs().max()))
    err = maxdiff(out_d, out_r)
    worst = sorted(res["grad_err"])[-5:]
    _log("%s: logits max|diff| %.2e (scale %.2f), loss diff %.2e, arg-max flips %d of %d (gap %.1e), activation-sign "
         "flips %d of %d (|z| <= %.1e), worst routed gradients %s"
         % (what, err, scale, abs(float(res["loss_d"]) - float(res["loss_r"])), res["flips"], res["entries"],
            res["max_gap"], res["sign_flips"], res["sign_entries"], res["sign_max_abs"], ["%.1e %s" % e for e in worst]))
    band = LOGIT_TOL * scale
    if "out_64" in res:
        d_gpu, d_cpu = adjudicate(res)
        r_gpu, r_cpu = adjudicate(res, rms=True)
        _log("%s: fp64 adjudication: |gpu - fp64| %.2e, |cpu_fp32 - fp64| %.2e (ratio %.2f; rms %.2e vs %.2e, ratio %.2f), "
             "1e-4 x scale = %.2e" % (what, d_gpu, d_cpu, d_gpu / max(d_cpu, 1e-30), r_gpu, r_cpu, r_gpu / max(r_cpu, 1e-30), band))
        # (rounds 3-5 carried a second, rms form for max-norm ratios between 1.5 and 1.75; round 5 needed it once -- 1.53 on the
        # full-width KITTI network at 49 652 points -- and round 6's A/B (profiles/r06_parity_split_ab.txt) found the cause: the tail
        # split of the paired GEMM, +-0 ms on the step.  The split is off by default now, the case is back at 1.13, the hatch is gone.)
        assert d_gpu <= max(band, ADJ_RATIO * d_cpu) and d_gpu <= band + d_cpu, (d_gpu, d_cpu, r_gpu, r_cpu, band)
        band = max(band, d_gpu + d_cpu)               # what the two fp32 evaluations may then differ by
    assert err <= band, (err, band)
    assert abs(float(res["loss_d"]) - float(res["loss_r"])) < 1e-5
    assert res["flips"] <= 4 + MAX_FLIP_RATE * res["entries"], (res["flips"], res["entries"])
    assert res["max_gap"] <= band, res["max_gap"]       # a flipped entry really was a tie (within the forward difference)
    # ReLU / LeakyReLU kinks: the oracle would have taken the other slope only where |z| is within the forward difference
    assert res["sign_flips"] <= 4 + MAX_FLIP_RATE * res["sign_entries"], (res["sign_flips"], res["sign_entries"])
    assert res["sign_max_abs"] <= band, res["sign_max_abs"]
    adj = {n: (a, b) for a, b, n in res.get("grad_adj", [])}
    if adj:
        wa = sorted(adj.items(), key=lambda kv: kv[1][0])[-3:]
        _log("%s: routed gradients vs fp64 (relative, gpu / cpu_fp32): %s"
             % (what, ["%.1e / %.1e %s" % (a, b, n) for n, (a, b) in wa]))
    for e, n in res["grad_err"]:
        if e <= GRAD_TOL:
            continue
        assert n in adj, (e, n)
        g_gpu, g_cpu = adj[n]
        assert g_gpu <= ADJ_RATIO * g_cpu + GRAD_TOL / 3, (n, e, g_gpu, g_cpu)


@pytest.mark.parametrize("ids,n_curves", [([0], 96), ([1, 2], 64)])
def test_model_forward_backward_matches_oracle(ids, n_curves):
    from curvecloudnet_amd.synth import make_batch
    cfg = hotpath_config(width=0.25)
    ref, mine = build_pair(cfg, in_dim=4, n_out=7)
    mine = mine.to(DEV)
    data = make_batch(ids, n_curves=n_curves)
    y = _labels(data.pos.size(0), 7, 3)
    ref.train(); mine.train()
    # The masked max of the SGCNN steps is not differentiable where two slots tie to the last bit: a CPU/GPU rounding
    # difference of 1e-7 can move ONE argmax, which re-routes that entry's gradient.  routed_parity counts such flips
    # (GPU table vs the oracle's own choice) and differentiates the oracle along the GPU's routes, so every gradient
    # tensor is held to GRAD_TOL.
    _check_routed(routed_parity(ref, mine, data, y, DEV), "hot path x0.25")
    for (n, br), (_, bd) in zip(ref.named_buffers(), mine.named_buffers()):
        assert maxdiff(bd.float(), br.float()) < 1e-4, n
    # eval mode uses the running statistics
    ref.eval(); mine.eval()
    torch.manual_seed(6)
    e_r = ref(data)
    torch.manual_seed(6)
    e_d = mine(batch_to(data, DEV))
    assert maxdiff(e_d, e_r) < LOGIT_TOL * max(1.0, float(e_r.abs().max()))


def test_full_width_hotpath_cloud_matches_oracle():
    """The section-8(a) hot-path network at FULL width on one BASELINE-size cloud (2048 curves, 49 652 points) against
    the CPU oracle: logits to 1e-4, every gradient tensor along identical arg-max routes."""
    from curvecloudnet_amd.synth import make_batch
    ref, mine = build_pair(hotpath_config(width=1.0), in_dim=4, n_out=20)
    mine = mine.to(DEV)
    data = make_batch([0])
    assert data.pos.size(0) == 49652
    y = _labels(data.pos.size(0), 20, 3)
    ref.train(); mine.train()
    _check_routed(routed_parity(ref, mine, data, y, DEV), "hot path x1.0, 49652 points")


def check_full_width_kitti_cloud_forward():
    """(Not collected in fp32 mode since round 5: test_full_width_kitti_backward_on_the_full_size_cloud below runs the same input
    through the same forward adjudication and the backward pass on top; tests/test_gpu_gemm_x3.py runs this forward-only form
    in bf16x3 mode.)  The reference's complete KITTI model section at full width (28.8 M parameters, the benchmark's network) on one
    BASELINE-size cloud (49 652 points): logits against the CPU oracle, adjudicated by the oracle evaluated in fp64 along
    the same routes.  33 steps / ~70 GEMM layers deep, contractions up to K = 3072, BatchNorm over a few hundred rows at
    the coarse levels: two fp32 evaluations of THIS network differ by more than 1e-4 from each other (the fp32 CPU oracle is
    itself several 1e-4 from the fp64 value), so the bound is the GPU's distance to the fp64 value against the fp32
    oracle's (see _check_routed)."""
    from curvecloudnet_amd.configs import kitti_config
    from curvecloudnet_amd.synth import make_batch
    ref, mine = build_pair(kitti_config(width=1.0), in_dim=4, n_out=20)
    mine = mine.to(DEV)
    data = make_batch([0])
    assert data.pos.size(0) == 49652
    y = _labels(data.pos.size(0), 20, 3)
    ref.train(); mine.train()
    with torch.no_grad():
        res = routed_parity(ref, mine, data, y, DEV, backward=False, fp64=True)
    _check_routed(res, "KITTI x1.0, 49652 points, forward")


def test_full_width_kitti_backward_matches_oracle():
    """The full-width KITTI network WITH backward on a 512-curve cloud (~12 k points; the oracle's forward + backward
    takes ~10 s there): logits, loss and every one of the 28.8 M-parameter model's gradient tensors along identical
    routes, with the fp64 evaluation alongside."""
    from curvecloudnet_amd.configs import kitti_config
    from curvecloudnet_amd.synth import make_batch
    ref, mine = build_pair(kitti_config(width=1.0), in_dim=4, n_out=20)
    mine = mine.to(DEV)
    data = make_batch([0], n_curves=512)
    y = _labels(data.pos.size(0), 20, 3)
    ref.train(); mine.train()
    _check_routed(routed_parity(ref, mine, data, y, DEV, fp64=True), "KITTI x1.0, %d points, forward + backward" % data.pos.size(0))


@pytest.mark.slow
def test_full_width_kitti_backward_on_the_full_size_cloud():
    """VERDICT r3 #2c: the full-width KITTI network WITH backward on the BASELINE-size cloud (2048 curves, 49 652 points),
    fp64 evaluation alongside: logits, loss and every gradient tensor of the 28.8 M-parameter model along identical routes.
    (The oracle's three passes take a few minutes of host time: once per suite.)"""
    from curvecloudnet_amd.configs import kitti_config
    from curvecloudnet_amd.synth import make_batch
    ref, mine = build_pair(kitti_config(width=1.0), in_dim=4, n_out=20)
    mine = mine.to(DEV)
    data = make_batch([0])
    assert data.pos.size(0) == 49652
    y = _labels(data.pos.size(0), 20, 3)
    ref.train(); mine.train()
    _check_routed(routed_parity(ref, mine, data, y, DEV, fp64=True), "KITTI x1.0, 49652 points, forward + backward")


@pytest.mark.parametrize("kortx", [False, True])
def test_full_width_shapenet_seg_and_kortx_on_2048_point_clouds(kortx):
    """BASELINE configs[0] at its own width (VERDICT r3 #2a): the reference's ShapeNet-seg section (and the Kortx section,
    the network configs[1] names) at FULL width on 2048-point clouds in the unit ball (85 curves, spacing ~0.012).
    Training mode with backward needs two clouds -- the category head's BatchNorm sees one row per cloud and torch refuses
    a single row in training mode -- routed, fp64-adjudicated; then configs[0] literally: ONE 2048-point cloud, eval-mode
    forward (the reference's CPU-runnable case) with the running statistics that step left behind."""
    from curvecloudnet_amd import configs
    from curvecloudnet_amd.synth import make_batch
    n_out = 10 if kortx else 50
    ref, mine = build_pair(configs.shapenet_seg_config(1.0, kortx=kortx), in_dim=3, n_out=n_out)
    mine = mine.to(DEV)

    def cloud(ids):
        d = make_batch(ids, n_curves=85, step=0.036)
        d.x = None
        d.pos = d.pos / 3.0
        return d
    two = cloud([0, 1])
    assert 3600 < two.pos.size(0) < 4600
    ref.train(); mine.train()
    what = "%s x1.0, 2 x ~2048 points" % ("Kortx" if kortx else "ShapeNet-seg")
    res = routed_parity(ref, mine, two, _labels(two.pos.size(0), n_out, 3), DEV, fp64=True,
                        fwd_kwargs={"shapenet-categories": torch.tensor([3, 11])})
    _check_routed(res, what + ", forward + backward")
    one = cloud([2])
    ref.eval(); mine.eval()
    with torch.no_grad():
        res = routed_parity(ref, mine, one, _labels(one.pos.size(0), n_out, 4), DEV, fp64=True, backward=False,
                            fwd_kwargs={"shapenet-categories": torch.tensor([7])})
    assert res["out_d"].shape == (one.pos.size(0), n_out)
    _check_routed(res, what.replace("2 x ~2048", "1 x %d" % one.pos.size(0)) + ", eval forward (configs[0])")


def test_full_size_cloud_properties():
    """One BASELINE-size cloud through the full-width hot path: finite outputs, per-row
    determinism of the integer stages, gradient w.r.t. every parameter."""
    from curvecloudnet_amd import ops
    from curvecloudnet_amd.model import ModelBase, segmentation_loss
    from curvecloudnet_amd.synth import make_batch
    cfg = {k: v for k, v in hotpath_config(width=1.0).items() if k != "type"}
    torch.manual_seed(0)
    model = ModelBase(4, 20, **cfg).to(DEV).train()
    data = batch_to(make_batch([0]), DEV)
    n = data.pos.size(0)
    assert n == 49652
    topo = ops.CurveTopology(data.batch, data.curve_idxs)
    assert topo.num_curves == 2048
    # CurveFPS: sorted, unique, keeps every curve start; radius groups stay on their curve
    idx = ops.curve_fps(data.pos, topo, 0.007, 0.5)
    assert bool((idx[1:] > idx[:-1]).all())
    assert bool(torch.isin(topo.curve_ptr[:-1].long(), idx).all())
    e = ops.radius_1d_group_subset(data.pos, idx, topo, 0.02)
    assert torch.equal(topo.cid[idx[e.row]], topo.cid[e.col])
    torch.manual_seed(1)
    out = model(data)
    assert out.shape == (n, 20) and bool(torch.isfinite(out).all())
    loss = segmentation_loss(out, _labels(n, 20, 1).to(DEV))
    loss.backward()
    for name, p in model.named_parameters():
        assert p.grad is not None and bool(torch.isfinite(p.grad).all()), name
    torch.manual_seed(1)
    again = model(data)
    assert maxdiff(again, out) < 1e-5      # the same draw gives the same forward (BN stats are order-independent)


def test_deferred_activations_give_the_same_bits():
    """ops.LAZY_ACT: hidden MLP activations that only feed the next Linear are applied inside that Linear's GEMMs
    (ccn_gemm_nt_xf / ccn_gemm_tn_ws_xf) instead of being written.  Logits, loss, running statistics and eval output
    are the same bits (a ReLU zero may be -0.0, which compares equal) as with every activation written out; gradients
    (which pass through atomic accumulations and are not run-to-run reproducible in either mode) differ by no more than
    two passes in the SAME mode do."""
    from curvecloudnet_amd import ops
    from curvecloudnet_amd.configs import kitti_config
    from curvecloudnet_amd.model import ModelBase, segmentation_loss
    from curvecloudnet_amd.synth import make_batch
    cfg = {k: v for k, v in kitti_config(width=1.0).items() if k != "type"}
    data = batch_to(make_batch([0, 1]), DEV)
    labels = _labels(data.pos.size(0), 20, 1).to(DEV)
    torch.manual_seed(0)
    model = ModelBase(4, 20, **cfg).to(DEV)
    state = {k: v.clone() for k, v in model.state_dict().items()}
    runs = []
    for lazy in (False, False, True):
        ops.LAZY_ACT = lazy
        ops.LAZY_ACT_COUNT.update(fused=0, written=0)
        try:
            model.load_state_dict(state)
            model.train()
            model.zero_grad(set_to_none=True)
            torch.manual_seed(1)
            out = model(data)
            loss = segmentation_loss(out, labels)
            loss.backward()
            grads = {n: p.grad.clone() for n, p in model.named_parameters()}
            buffers = {n: b.clone() for n, b in model.named_buffers()}
            model.eval()
            torch.manual_seed(1)
            with torch.no_grad():
                ev = model(data)
            runs.append((out.detach().clone(), loss.detach().clone(), grads, buffers, ev.clone(), dict(ops.LAZY_ACT_COUNT)))
        finally:
            ops.LAZY_ACT = True
    (o0, l0, g0, b0, e0, c0), (oa, la, ga, ba, ea, ca), (o1, l1, g1, b1, e1, c1) = runs
    assert c0 == {"fused": 0, "written": 0}
    assert c1["fused"] >= 10, c1               # the fused kernels really ran (train + eval passes)
    assert torch.equal(o0, o1) and torch.equal(l0, l1) and torch.equal(e0, e1)
    for n in b0:
        assert torch.equal(b0[n], b1[n]), "buffer %s differs" % n
    gscale = max(float(g.abs().max()) for g in g0.values())
    for n in g0:
        scale = float(g0[n].abs().max())
        same_mode = float((g0[n] - ga[n]).abs().max())
        across = float((g0[n] - g1[n]).abs().max())
        # (same_mode is ONE sample of the atomic-order noise and can come out as zero for a tensor -- since round 5, with the
        # first edge layers' backward free of atomics, it is zero for most: the floor is five times the typical noise, 1e-5 of
        # the tensor's scale, plus 1e-5 of the largest gradient for the tensors whose gradient is zero in exact arithmetic --
        # a bias ahead of a BatchNorm: max |g| ~ 1e-7, rounding residue of a cancelling sum that the two modes' weight-
        # gradient kernels add up in different orders; a wrong transform moves gradients by 1e-2 of their scale and more)
        assert across <= 3.0 * same_mode + 5e-5 * scale + 1e-5 * gscale, (n, across, same_mode, scale, gscale)


def test_full_kitti_config_matches_oracle():
    """The complete KITTI / nuScenes step list (33 steps: sa-geo, voxel and farthest-point SA levels, FP
    up-sampling, 10 SGCNN layers) at 1/8 width: logits and gradients against the CPU oracle."""
    from oracle import torch_ref as R
    from curvecloudnet_amd.configs import kitti_config
    from curvecloudnet_amd.model import segmentation_loss
    from curvecloudnet_amd.synth import make_batch
    ref, mine = build_pair(kitti_config(width=0.125), in_dim=4, n_out=20)
    mine = mine.to(DEV)
    data = make_batch([0, 1], n_curves=200)
    y = _labels(data.pos.size(0), 20, 3)
    ref.train(); mine.train()
    # 11 max aggregations (10 SGCNN levels + the max SA level): flips counted, gradients compared along the GPU's routes
    _check_routed(routed_parity(ref, mine, data, y, DEV), "KITTI x0.125")


def test_shapenet_seg_config_matches_oracle():
    """The ShapeNet-seg / Kortx step list (x=None input, biased MLPs, v1 attention widths, conv1d-fast-v1, ball-query SA
    with farthest point sampling, sparse exact-kNN SGCNN, FP, category one-hot head) at 1/8 width."""
    from oracle import torch_ref as R
    from curvecloudnet_amd.configs import shapenet_seg_config
    from curvecloudnet_amd.model import segmentation_loss
    from curvecloudnet_amd.synth import make_batch
    ref, mine = build_pair(shapenet_seg_config(width=0.125), in_dim=3, n_out=50)
    mine = mine.to(DEV)
    data = make_batch([0, 1], n_curves=90)
    data.x = None
    data.pos = data.pos / 3.0                       # ShapeNet clouds live in the unit ball
    cats = torch.tensor([3, 11])
    y = _labels(data.pos.size(0), 50, 3)
    ref.train(); mine.train()
    res = routed_parity(ref, mine, data, y, DEV, fwd_kwargs={"shapenet-categories": cats})
    assert res["out_d"].shape == (data.pos.size(0), 50)
    _check_routed(res, "ShapeNet-seg x0.125")


@pytest.mark.parametrize("which", ["a2d2", "shapenet-cls", "kortx"])
def test_remaining_reference_configs_match_oracle(which):
    """A2D2 (FRNN + attention in sparse SGCNN), ShapeNet classification (global pooling head) and Kortx (k=7
    convolutions, K=30; the network BASELINE configs[1] names) at 1/8 width: logits, loss and every gradient tensor
    against the CPU oracle along identical routes, as for the KITTI section."""
    from curvecloudnet_amd import configs
    from curvecloudnet_amd.synth import make_batch
    # (classification: the head's BatchNorm layers see ONE row per cloud.  Over two rows a BatchNorm output is +-gamma + beta
    # whatever the input, so every gradient in front of it is exactly zero and both sides would be compared on rounding
    # noise alone: eight clouds there.)
    data = make_batch(list(range(8)), n_curves=24) if which == "shapenet-cls" else make_batch([0, 1], n_curves=90)
    if which == "a2d2":
        cfg, in_dim, n_out = configs.a2d2_config(0.125), 4, 12
    elif which == "kortx":
        cfg, in_dim, n_out = configs.shapenet_seg_config(0.125, kortx=True), 3, 10
        data.x = None
    else:
        cfg, in_dim, n_out = configs.shapenet_cls_config(0.125), 3, 16
        data.x = None
    if in_dim == 3:
        data.pos = data.pos / 3.0
    ref, mine = build_pair(cfg, in_dim=in_dim, n_out=n_out)
    mine = mine.to(DEV)
    ref.train(); mine.train()
    kw = {"shapenet-categories": torch.tensor([1, 2])} if which == "kortx" else {}
    n_rows = 8 if which == "shapenet-cls" else data.pos.size(0)
    y = _labels(n_rows, n_out, 3)
    res = routed_parity(ref, mine, data, y, DEV, fwd_kwargs=kw, fp64=True)
    assert res["out_r"].shape == (n_rows, n_out)
    _check_routed(res, "%s x0.125" % which)


def test_geometry_stream_modes_agree(monkeypatch):
    """The position-only work runs on a side stream (steps.ForwardContext.geometry), either step by step inside
    forward or ahead of it (ModelBase.prepare).  Runs of the same model and input -- side stream off, on, on with the
    side stream artificially delayed at every block, and prepared ahead with the delay -- must give bit-identical logits (the forward has no atomics) and the same gradients up to the
    atomic-add order of the weight-gradient kernels.  A missing stream dependency fails the stress run."""
    from curvecloudnet_amd import configs
    from curvecloudnet_amd.model import build_model, segmentation_loss
    from curvecloudnet_amd.synth import make_batch
    torch.manual_seed(2)
    model = build_model(configs.kitti_config(width=0.125), in_dim=4, n_out=20).to(DEV).train()
    data = batch_to(make_batch([3, 4], n_curves=160), DEV)
    y = _labels(data.pos.size(0), 20, 8).to(DEV)
    runs = {}
    for mode in ("0", "1", "stress", "stress", "prepared", "async"):
        monkeypatch.setenv("CCN_GEOMETRY_STREAM", "stress" if mode in ("prepared", "async") else mode)
        model.zero_grad(set_to_none=True)
        torch.manual_seed(11)
        if mode == "prepared":       # ModelBase.prepare: all position-only work first, features afterwards
            plan = model.prepare(data)
            assert plan is not None
            out = model(data, plan=plan)
        elif mode == "async":        # ... driven by the worker thread (ModelBase.prepare_async), seeded there
            plan = model.prepare_async(data, seed=11).result()
            assert plan is not None
            out = model(data, plan=plan)
        else:
            out = model(data)
        segmentation_loss(out, y).backward()
        torch.cuda.synchronize()
        grads = torch.cat([p.grad.flatten() for p in model.parameters() if p.grad is not None])
        assert torch.isfinite(out).all()
        key = mode if mode not in runs else mode + "2"
        runs[key] = (out.detach().clone(), grads.clone())
    base_out, base_grad = runs["0"]
    for mode, (out, grads) in runs.items():
        assert torch.equal(out, base_out), "logits differ between CCN_GEOMETRY_STREAM=0 and %s" % mode
        rel = float((grads - base_grad).norm() / base_grad.norm())
        assert rel < 1e-4, (mode, rel)


def test_fused_weight_gradient_accumulation_matches_autograd():
    """With a GradientAllReduce attached, the HIP layers add weight gradients straight into the bucket views (no tensor
    is handed to autograd for them).  Gradients must equal the plain autograd result, also when two backward passes
    accumulate, and every parameter's bucket slot must be filled."""
    import copy
    from curvecloudnet_amd import configs
    from curvecloudnet_amd.model import build_model, segmentation_loss
    from curvecloudnet_amd.parallel import GradientAllReduce
    from curvecloudnet_amd.synth import make_batch
    torch.manual_seed(2)
    plain = build_model(configs.kitti_config(width=0.125), in_dim=4, n_out=20).to(DEV).train()
    fused = copy.deepcopy(plain)
    sync = GradientAllReduce(fused)
    data = batch_to(make_batch([3, 4], n_curves=120), DEV)
    y = _labels(data.pos.size(0), 20, 8).to(DEV)
    for model in (plain, fused):
        for rep in range(2):                        # accumulate over two passes
            torch.manual_seed(11)
            segmentation_loss(model(data), y).backward()
    sync.finish()
    ga, gb = [], []
    for (n, a), (_, b) in zip(plain.named_parameters(), fused.named_parameters()):
        assert b.grad is not None and b.grad.data_ptr() == b._ccn_main_grad.data_ptr(), n
        assert float(b.grad.abs().max()) > 0 or float(a.grad.abs().max()) == 0, n
        ga.append(a.grad.flatten())
        gb.append(b.grad.flatten())
    ga, gb = torch.cat(ga), torch.cat(gb)
    rel = float((ga - gb).norm() / ga.norm())
    assert rel < 1e-4, rel               # atomic accumulation order only (same bound as between two plain runs)


# ---------------------------------------------------------------- BASELINE configs[4]: A2D2 section, mixed curve lengths
def _mixed_batch():
    """Two clouds whose curves span the whole mixed-length range of configs[4]: single-point curves, 2-point curves,
    curves shorter than the conv kernel, and 512-point curves, in one batch; a cloud ending in a single-point curve."""
    from curvecloudnet_amd.synth import make_batch, make_cloud
    from types import SimpleNamespace
    lens = [[1, 512, 3, 1, 1, 17, 200, 2, 64, 5, 1, 512, 33, 1], [512, 1, 1, 4, 90, 2, 1, 300, 7, 1]]
    clouds = [make_cloud(10 + i, lengths=l) for i, l in enumerate(lens)]
    return SimpleNamespace(
        x=torch.cat([c.x for c in clouds]), pos=torch.cat([c.pos for c in clouds]),
        curve_idxs=torch.cat([c.curve_idxs for c in clouds]),
        batch=torch.cat([torch.full((c.pos.size(0),), i, dtype=torch.long) for i, c in enumerate(clouds)]),
        num_clouds=len(clouds))


def test_a2d2_section_mixed_curve_lengths_matches_oracle():
    """BASELINE configs[4]: the reference's A2D2 model section (FRNN + attention aggregation in the sparse SGCNN levels,
    conv1d-fast-v1) on clouds with MIXED curve lengths -- length-1 and length-512 curves in one batch -- logits and
    routed gradients against the CPU oracle at 1/8 width."""
    from curvecloudnet_amd import configs
    ref, mine = build_pair(configs.a2d2_config(0.125), in_dim=4, n_out=12)
    mine = mine.to(DEV)
    data = _mixed_batch()
    assert int(torch.bincount(data.curve_idxs[data.batch == 0]).max()) == 512
    y = _labels(data.pos.size(0), 12, 3)
    ref.train(); mine.train()
    _check_routed(routed_parity(ref, mine, data, y, DEV), "A2D2 x0.125, mixed lengths")


def test_mixed_length_synthetic_clouds_full_width_properties():
    """configs[4] at the benchmark's own shape (log-normal curve lengths clamped to [1, 512], full-width A2D2 section):
    curve segment offsets bit-exact against a host recomputation, finite logits / gradients, repeatable forward."""
    from curvecloudnet_amd import configs, ops
    from curvecloudnet_amd.model import build_model, segmentation_loss
    from curvecloudnet_amd.synth import make_batch
    cpu = make_batch([0, 1], n_curves=1024, mixed_lengths=True)
    data = batch_to(cpu, DEV)
    topo = ops.CurveTopology(data.batch, data.curve_idxs)
    # segment offsets: starts of the runs of (cloud, curve) on the host
    key = cpu.batch * (1 << 20) + cpu.curve_idxs
    starts = torch.cat([torch.zeros(1, dtype=torch.long), torch.nonzero(key[1:] != key[:-1]).flatten() + 1,
                        torch.tensor([key.numel()])])
    assert torch.equal(topo.curve_ptr.cpu().long(), starts)
    lens = starts[1:] - starts[:-1]
    assert int(lens.min()) == 1 and int(lens.max()) > 100
    torch.manual_seed(0)
    model = build_model(configs.a2d2_config(1.0), in_dim=4, n_out=55).to(DEV).train()
    torch.manual_seed(1)
    out = model(data)
    assert out.shape == (cpu.pos.size(0), 55) and bool(torch.isfinite(out).all())
    segmentation_loss(out, _labels(cpu.pos.size(0), 55, 1).to(DEV)).backward()
    for name, p in model.named_parameters():
        assert p.grad is not None and bool(torch.isfinite(p.grad).all()), name
    torch.manual_seed(1)
    assert maxdiff(model(data), out) < 1e-5


def test_a2d2_fp16_full_width_mixed_length_clouds_and_graph_replay():
    """BASELINE configs[4] at its own size in its own arithmetic (VERDICT r3 #2b): 8 mixed-length clouds (2048 log-normal
    curves each, lengths 1..512), the full-width A2D2 section, fp16 features.  Finite logits and gradients for every
    parameter; NO fp16 overflow -- every stored fp16 activation row of the forward stays below 6e4 in magnitude (512 / 1024
    channels are where a range problem would show); the features of the first steps within the 16-bit band of the fp32
    path's; and ``graph.CapturedForward.replay()`` bit-identical to the eager pass at full width."""
    from curvecloudnet_amd import configs, ops
    from curvecloudnet_amd.graph import CapturedForward
    from curvecloudnet_amd.model import build_model, segmentation_loss
    from curvecloudnet_amd.synth import make_batch
    cpu = make_batch(list(range(8)), n_curves=2048, mixed_lengths=True)
    data = batch_to(cpu, DEV)
    n = cpu.pos.size(0)
    assert n > 8 * 40000
    labels = _labels(n, 55, 1).to(DEV)
    torch.manual_seed(0)
    model = build_model(configs.a2d2_config(1.0), in_dim=4, n_out=55).to(DEV).train()
    state = {k: v.clone() for k, v in model.state_dict().items()}
    watch = (0, 1, 2, 3)                      # sa-geo, mlp, conv1d-fast-v1, skip-connect
    feats, outs, losses = {}, {}, {}
    hooks = [model.steps[i].register_forward_hook(
        lambda mod, args, out, i=i: feats.setdefault(i, []).append(out[0].detach().float().clone())) for i in watch]
    peak = {}
    try:
        for mode in ("fp32", "fp16"):
            ops.set_mlp_dtype(mode)
            ops.ROWS16_WATCH = [] if mode == "fp16" else None
            try:
                model.load_state_dict(state)
                model.zero_grad(set_to_none=True)
                torch.manual_seed(1)
                out = model(data)
                assert out.shape == (n, 55) and bool(torch.isfinite(out).all())
                loss = segmentation_loss(out, labels)
                losses[mode] = float(loss)
                if mode == "fp16":
                    stored = [r for r in ops.ROWS16_WATCH if r.dtype == torch.float16 and r.numel()]
                    assert len(stored) >= 20, len(stored)             # the fp16 storage path really ran
                    for r in stored:
                        big = float(r.float().abs().max())
                        assert big < 6e4, "fp16 activation of %s reaches %g" % (tuple(r.shape), big)
                        peak[r.size(1)] = max(peak.get(r.size(1), 0.0), big)
                    ops.ROWS16_WATCH = None
                    loss.backward()
                    for name, p in model.named_parameters():
                        assert p.grad is not None and bool(torch.isfinite(p.grad).all()), name
                outs[mode] = out.detach().clone()
                del out, loss
            finally:
                ops.ROWS16_WATCH = None
                ops.set_mlp_dtype("fp32")
    finally:
        for h in hooks:
            h.remove()

    def rel(a, b):
        return float((a - b).norm() / b.norm())
    per_step = {i: rel(feats[i][1], feats[i][0]) for i in watch}
    _log("A2D2 x1.0, 8 mixed-length clouds (%d points), fp16 path vs fp32 path: relative l2 of the features after steps %s = %s, of "
         "the logits %.2e; loss %.5f vs %.5f; largest stored fp16 activation by width: %s"
         % (n, list(watch), ["%.1e" % per_step[i] for i in watch], rel(outs["fp16"], outs["fp32"]), losses["fp16"],
            losses["fp32"], {k: "%.3g" % v for k, v in sorted(peak.items())}))
    assert all(v > 1e-6 for v in per_step.values()), per_step          # a different arithmetic really ran
    assert per_step[0] < 5e-3 and per_step[1] < 1e-2 and per_step[2] < 2e-2 and per_step[3] < 3e-2, per_step
    assert abs(losses["fp16"] - losses["fp32"]) < 0.05 * losses["fp32"], losses
    # the whole-width graph: the inference forward of the same batch, fp16 features, replayed
    ops.set_mlp_dtype("fp16")
    try:
        model.load_state_dict(state)
        model.eval()
        torch.manual_seed(9)
        cap = CapturedForward(model, data)
        eager = cap.eager().clone()
        first = cap.replay().clone()
        second = cap.replay().clone()
        torch.cuda.synchronize()
        assert bool(torch.isfinite(eager).all())
        assert torch.equal(first, eager) and torch.equal(second, first)
    finally:
        ops.set_mlp_dtype("fp32")


# ---------------------------------------------------------------- BASELINE configs[3]: 4 x ~120k-point clouds, KITTI section
def test_kitti_4x120k_properties():
    """configs[3] per-GPU shape: 4 clouds of 4900 curves (~120k points each) through the full-width KITTI section.
    Size-independent properties: curve offsets, FRNN result sorted / inside the radius / equal to an exhaustive search
    on sampled queries, finite logits and gradients for every parameter, repeatable forward."""
    from curvecloudnet_amd import configs, ops
    from curvecloudnet_amd.model import build_model, segmentation_loss
    from curvecloudnet_amd.synth import make_batch
    cpu = make_batch([0, 1, 2, 3], n_curves=4900)
    data = batch_to(cpu, DEV)
    n = cpu.pos.size(0)
    assert n > 4 * 110000
    topo = ops.CurveTopology(data.batch, data.curve_idxs)
    assert topo.num_curves == 4 * 4900 and topo.num_clouds == 4
    assert bool((topo.curve_ptr[1:] > topo.curve_ptr[:-1]).all()) and int(topo.curve_ptr[-1]) == n
    # FRNN at this size (K = 20, r = 0.04: the first SGCNN level's search), checked on 512 sampled queries per cloud
    padded, _ = ops.to_batch_padded(data.pos, topo)
    K, r = 20, 0.04
    idx, d2 = ops.fast_knn(padded, padded, topo.lengths, topo.lengths, K, r, return_dists=True)
    gen = torch.Generator().manual_seed(0)
    for b in range(4):
        ln = int(topo.lengths[b])
        q = torch.randint(0, ln, (512,), generator=gen).to(DEV)
        pts = padded[b, :ln]
        diff = pts[q][:, None, :] - pts[None, :, :]
        # the kernel's own expression: fma(dz, dz, fma(dy, dy, dx*dx)) -- evaluated in fp64 here and compared by SET where
        # distances are distinct at fp32 resolution
        full = (diff.double() ** 2).sum(-1)
        inside = full < float(torch.tensor(r, dtype=torch.float32)) ** 2
        got = idx[b, q]
        cnt = (got >= 0).sum(1)
        want_cnt = inside.sum(1).clamp(max=K)
        assert bool((cnt - want_cnt).abs().max() <= 1)            # (a point exactly on the radius may round either way)
        dd = d2[b, q]
        valid = got >= 0
        assert bool((dd[valid] < r * r * (1 + 1e-6)).all())
        srt = torch.where(valid, dd, torch.full_like(dd, float("inf")))
        assert bool((srt[:, 1:] >= srt[:, :-1]).all())             # ascending, padding last
        # the K-th neighbour distance equals the exhaustive K-th smallest
        kth = torch.sort(torch.where(inside, full, torch.full_like(full, float("inf"))), dim=1)[0][:, :K]
        ref_d = torch.where(torch.isfinite(kth), kth, torch.full_like(kth, float("inf"))).float()
        assert float((torch.where(valid, dd, torch.full_like(dd, float("inf"))) - ref_d).nan_to_num(0, 0, 0).abs().max()) < 1e-9
    del idx, d2, padded
    torch.manual_seed(0)
    model = build_model(configs.kitti_config(1.0), in_dim=4, n_out=20).to(DEV).train()
    torch.manual_seed(1)
    out = model(data)
    assert out.shape == (n, 20) and bool(torch.isfinite(out).all())
    segmentation_loss(out, _labels(n, 20, 1).to(DEV)).backward()
    for name, p in model.named_parameters():
        assert p.grad is not None and bool(torch.isfinite(p.grad).all()), name
    torch.manual_seed(1)
    assert maxdiff(model(data), out) < 1e-5


# ---------------------------------------------------------------- BASELINE configs[2]: 16 x ~35k-point clouds, nuScenes section, bf16
def test_nuscenes_16x35k_bf16_properties():
    """configs[2] at its own size: 16 clouds of 1430 curves (~35k points each) through the full-width nuScenes section
    with the 16-bit MLP path (``ops.set_mlp_dtype("bf16")``).  Size-independent properties: curve offsets exact, finite
    logits and a finite gradient for every parameter, a repeatable forward, BatchNorm statistics finite and updated, the
    16-bit kernels really ran (features differ from the fp32 path's), and the features of the FIRST steps -- before the
    33-step network in training mode has amplified the perturbation (a random-initialised network of this depth turns the
    fp32 rounding noise of 6e-8 into 2e-4 at the logits, see the fp64 adjudication above, and saturates on bf16's 4e-3) --
    stay within the 16-bit band of the fp32 path's (operands rounded to 8 significant bits, fp32 accumulation); the loss of
    the two paths agrees to a few per cent."""
    from curvecloudnet_amd import configs, ops
    from curvecloudnet_amd.model import build_model, segmentation_loss
    from curvecloudnet_amd.synth import make_batch
    cpu = make_batch(list(range(16)), n_curves=1430)
    data = batch_to(cpu, DEV)
    n = cpu.pos.size(0)
    assert 16 * 33000 < n < 16 * 37000
    topo = ops.CurveTopology(data.batch, data.curve_idxs)
    assert topo.num_curves == 16 * 1430 and topo.num_clouds == 16
    key = cpu.batch * (1 << 20) + cpu.curve_idxs
    starts = torch.cat([torch.zeros(1, dtype=torch.long), torch.nonzero(key[1:] != key[:-1]).flatten() + 1,
                        torch.tensor([key.numel()])])
    assert torch.equal(topo.curve_ptr.cpu().long(), starts)
    labels = _labels(n, 17, 1).to(DEV)
    torch.manual_seed(0)
    model = build_model(configs.nuscenes_config(1.0), in_dim=4, n_out=17).to(DEV).train()
    state = {k: v.clone() for k, v in model.state_dict().items()}
    watch = (0, 1, 2, 3, 4)                       # conv1d-fast-v2, sa-geo, mlp, sgcnn, skip-connect: the full-resolution levels
    feats, outs, losses = {}, {}, {}
    hooks = [model.steps[i].register_forward_hook(
        lambda mod, args, out, i=i: feats.setdefault(i, []).append(out[0].detach().clone())) for i in watch]
    try:
        for mode in ("fp32", "bf16"):
            ops.set_mlp_dtype(mode)
            try:
                model.load_state_dict(state)
                model.zero_grad(set_to_none=True)
                torch.manual_seed(1)
                out = model(data)
                assert out.shape == (n, 17) and bool(torch.isfinite(out).all())
                loss = segmentation_loss(out, labels)
                losses[mode] = float(loss)
                if mode == "bf16":
                    loss.backward()
                    for name, p in model.named_parameters():
                        assert p.grad is not None and bool(torch.isfinite(p.grad).all()), name
                    for name, b in model.named_buffers():
                        assert bool(torch.isfinite(b.float()).all()), name
                        if name.endswith("running_mean"):
                            assert not torch.equal(b, state[name]), name
                outs[mode] = out.detach().clone()
                del out, loss
            finally:
                ops.set_mlp_dtype("fp32")
    finally:
        for h in hooks:
            h.remove()
    ops.set_mlp_dtype("bf16")
    try:
        model.load_state_dict(state)
        torch.manual_seed(1)
        with torch.no_grad():
            assert maxdiff(model(data), outs["bf16"]) < 1e-5          # repeatable
    finally:
        ops.set_mlp_dtype("fp32")

    def rel(a, b):
        return float((a - b).norm() / b.norm())
    per_step = {i: rel(feats[i][1], feats[i][0]) for i in watch}
    r_out = rel(outs["bf16"], outs["fp32"])
    _log("nuScenes x1.0, 16 x ~35k points (%d), bf16 MLP path vs fp32 path: relative l2 of the features after steps %s = %s, "
         "of the logits %.2e; loss %.5f vs %.5f"
         % (n, list(watch), ["%.1e" % per_step[i] for i in watch], r_out, losses["bf16"], losses["fp32"]))
    assert all(v > 1e-5 for v in per_step.values()), per_step      # a different arithmetic really ran
    assert per_step[0] < 2e-2 and per_step[1] < 3e-2 and per_step[2] < 5e-2, per_step
    assert r_out < 1.0
    assert abs(losses["bf16"] - losses["fp32"]) < 0.05 * losses["fp32"], losses
