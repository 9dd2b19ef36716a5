"""GPU parity, integer results: HIP kernels (through the C-ABI) vs the CPU oracle and the golden
vectors made from the reference.  Everything here must be BIT-EXACT."""
import numpy as np
import pytest
import torch

from tests.util import CASES, golden, t

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ops():
    from curvecloudnet_amd import ops
    return ops


def _synth(ids, **kw):
    from curvecloudnet_amd.synth import make_batch
    return make_batch(ids, **kw)


# ---------------------------------------------------------------- A1 / A2
@pytest.mark.parametrize("case", CASES)
def test_segment_ptr_and_topology_golden(case):
    ops = _ops()
    g = golden("index_algebra")
    batch, p2c = t(g[case + ".batch"], DEV), t(g[case + ".p2c"], DEV)
    assert torch.equal(ops.batch2ptr(batch, with_ends=True).cpu(), t(g[case + ".cloud_ptr"]))
    assert torch.equal(ops.batch2ptr(batch).cpu(), t(g[case + ".cloud_ptr_interior"]))
    topo = ops.CurveTopology(batch, p2c)
    assert torch.equal(topo.glob.cpu(), t(g[case + ".glob"]))
    assert torch.equal(topo.curve_ptr.cpu().long(), t(g[case + ".curve_ptr"]))
    assert torch.equal(topo.cloud_ptr.cpu(), t(g[case + ".cloud_ptr"]))
    assert torch.equal(ops.batch2ptr(topo.glob).cpu(), t(g[case + ".curve_ptr_interior"]))
    assert torch.equal(ops.curveidx_local2global(p2c, batch).cpu(), t(g[case + ".glob"]))
    feats = t(g[case + ".feats"], DEV)
    padded, mask = ops.to_batch_padded(feats, topo)
    assert torch.equal(padded.cpu(), t(g[case + ".padded"]))
    assert torch.equal(mask.cpu(), t(g[case + ".mask"]))
    assert torch.equal(topo.lengths.cpu(), t(g[case + ".lengths"]))


def test_segment_ptr_rejects_unsorted_and_handles_edges():
    ops = _ops()
    with pytest.raises(AssertionError):
        ops.batch2ptr(torch.tensor([0, 1, 0, 2], device=DEV))
    one = ops.batch2ptr(torch.tensor([5], device=DEV), with_ends=True)
    assert one.tolist() == [0, 1]
    assert ops.batch2ptr(torch.tensor([3, 3, 3], device=DEV)).numel() == 0
    big = torch.sort(torch.randint(0, 5000, (300001,), generator=torch.Generator().manual_seed(1)))[0]
    from oracle import torch_ref as R
    assert torch.equal(ops.batch2ptr(big.to(DEV), with_ends=True).cpu(), R.segment_starts(big, True))
    with pytest.raises(AssertionError):
        ops.CurveTopology(torch.tensor([0, 0, 2, 2], device=DEV), torch.tensor([0, 1, 0, 1], device=DEV), 3)


def test_topology_large_matches_oracle():
    ops = _ops()
    from oracle import torch_ref as R
    d = _synth([3, 4, 5], n_curves=700)
    topo = ops.CurveTopology(d.batch.to(DEV), d.curve_idxs.to(DEV))
    glob = R.curve_ids_global(d.curve_idxs, d.batch)
    assert torch.equal(topo.glob.cpu(), glob)
    assert torch.equal(topo.curve_ptr.cpu().long(), R.segment_starts(glob, True))
    assert torch.equal(topo.cid.cpu().long(), glob)           # synthetic ids are dense
    assert topo.num_curves == 2100 and topo.num_clouds == 3


# ---------------------------------------------------------------- A7
def test_curve_fps_golden():
    ops = _ops()
    g = golden("curve_fps")
    keys = sorted({k.rsplit(".", 1)[0] for k in g.files})
    assert len(keys) == 8
    for key in keys:
        pos, batch, p2c = t(g[key + ".pos"], DEV), t(g[key + ".batch"], DEV), t(g[key + ".p2c"], DEV)
        topo = ops.CurveTopology(batch, p2c)
        idx = ops.curve_fps(pos, topo, float(g[key + ".spacing"]), float(g[key + ".u"][0]))
        assert torch.equal(idx.cpu(), t(g[key + ".idx"])), key


@pytest.mark.parametrize("ids,spacing", [([0], 0.007), ([1, 2, 3], 0.007), ([4, 5], 0.02), ([6], 0.03)])
def test_curve_fps_matches_oracle(ids, spacing):
    ops = _ops()
    from oracle import torch_ref as R
    d = _synth(ids, n_curves=600)
    u = torch.rand(1, generator=torch.Generator().manual_seed(9))
    want = R.curve_fps(d.pos, d.batch, d.curve_idxs, spacing, u)
    topo = ops.CurveTopology(d.batch.to(DEV), d.curve_idxs.to(DEV))
    got = ops.curve_fps(d.pos.to(DEV), topo, spacing, float(u))
    assert torch.equal(got.cpu(), want)


# ---------------------------------------------------------------- A8 / A9
@pytest.mark.parametrize("case", CASES)
def test_curve_groups_golden(case):
    ops = _ops()
    g = golden("curve_group")
    pos, batch, p2c = t(g[case + ".pos"], DEV), t(g[case + ".batch"], DEV), t(g[case + ".p2c"], DEV)
    idx = t(g[case + ".idx"], DEV)
    topo = ops.CurveTopology(batch, p2c)
    for radius in (0.02, 0.006):
        e = ops.radius_1d_group_subset(pos, idx, topo, radius)
        key = "%s.r%g" % (case, radius)
        assert torch.equal(e.row.cpu(), t(g[key + ".row"])), key
        assert torch.equal(e.col.cpu(), t(g[key + ".col"])), key
        counts = torch.bincount(e.row, minlength=idx.numel())
        assert torch.equal((e.offsets[1:] - e.offsets[:-1]).long(), counts)
    for k in (3, 1):
        row, col = ops.knn_1d_group_superset(pos, idx, topo, k)
        assert torch.equal(row.cpu(), t(g["%s.k%d.row" % (case, k)]))
        assert torch.equal(col.cpu(), t(g["%s.k%d.col" % (case, k)]))


@pytest.mark.parametrize("ids", [[0], [1, 2, 3]])
def test_curve_groups_match_oracle(ids):
    ops = _ops()
    from oracle import torch_ref as R
    d = _synth(ids, n_curves=500)
    u = torch.tensor([0.41])
    idx = R.curve_fps(d.pos, d.batch, d.curve_idxs, 0.007, u)
    topo = ops.CurveTopology(d.batch.to(DEV), d.curve_idxs.to(DEV))
    pos = d.pos.to(DEV)
    for radius in (0.02, 0.05):
        want = R.curve_radius_group(d.pos, idx, d.curve_idxs, d.batch, radius)
        e = ops.radius_1d_group_subset(pos, idx.to(DEV), topo, radius)
        assert torch.equal(e.row.cpu(), want[0]) and torch.equal(e.col.cpu(), want[1])
    want = R.curve_knn_superset(d.pos, idx, d.curve_idxs, d.batch, 3)
    row, col = ops.knn_1d_group_superset(pos, idx.to(DEV), topo, 3)
    assert torch.equal(row.cpu(), want[0]) and torch.equal(col.cpu(), want[1])


# ---------------------------------------------------------------- A11
@pytest.fixture(params=[1, 2, 3], ids=["thread-per-query", "team32", "team64"])
def frnn_mode(request):
    """Every FRNN query kernel: one thread per query, teams of 32 / 64 lanes per query (ccn_frnn_query_mode)."""
    from curvecloudnet_amd import _lib
    _lib.lib().ccn_frnn_query_mode(request.param)
    yield request.param
    _lib.lib().ccn_frnn_query_mode(0)


def _frnn_case(B, P1, P2, K, r, seed, ragged=True, same=False):
    gen = torch.Generator().manual_seed(seed)
    p2 = torch.rand(B, P2, 3, generator=gen) * torch.tensor([2.0, 2.0, 0.4])
    p1 = p2[:, :P1].clone() if same else torch.rand(B, P1, 3, generator=gen) * torch.tensor([2.0, 2.0, 0.4])
    if ragged:
        l1 = torch.randint(max(1, P1 // 2), P1 + 1, (B,), generator=gen)
        l2 = torch.randint(max(1, P2 // 2), P2 + 1, (B,), generator=gen)
        l1[0], l2[0] = P1, P2
    else:
        l1, l2 = torch.full((B,), P1), torch.full((B,), P2)
    if same:
        l1 = torch.minimum(l1, l2)
    return p1, p2, l1, l2


@pytest.mark.parametrize("B,P1,P2,K,r", [(1, 500, 500, 8, 0.15), (3, 257, 1000, 20, 0.2), (2, 1000, 300, 32, 0.5),
                                          (4, 64, 64, 5, 0.05), (1, 3000, 3000, 20, 0.08), (2, 100, 100, 40, 3.0)])
def test_frnn_bit_exact_vs_bruteforce(B, P1, P2, K, r, frnn_mode):
    ops = _ops()
    from oracle import torch_ref as R
    p1, p2, l1, l2 = _frnn_case(B, P1, P2, K, r, seed=B * 1000 + P1)
    want, want_d = R.frnn_bruteforce(p1, p2, l1, l2, K, r, return_dists=True)
    got, got_d = ops.fast_knn(p1.to(DEV), p2.to(DEV), l1.to(DEV), l2.to(DEV), K, r, return_dists=True)
    assert torch.equal(got.cpu(), want)
    assert torch.equal(got_d.cpu(), want_d)            # same fma chain => identical distances


def test_frnn_per_cloud_radius_empty_and_lattice(frnn_mode):
    ops = _ops()
    from oracle import torch_ref as R
    # per-cloud radii, one cloud with a single point, K larger than any neighbourhood
    p1, p2, l1, l2 = _frnn_case(3, 200, 200, 50, 0.1, seed=5, same=True)
    l1[1], l2[1] = 1, 1
    r = torch.tensor([0.05, 0.3, 0.12])
    want = R.frnn_bruteforce(p1, p2, l1, l2, 50, r)
    got = ops.fast_knn(p1.to(DEV), p2.to(DEV), l1.to(DEV), l2.to(DEV), 50, r)
    assert torch.equal(got.cpu(), want)
    assert (got[1, 1:] == -1).all() and got[1, 0, 0] == 0
    # lattice: many exactly equal distances -> ties resolved by index on both sides, r between shells
    ax = torch.arange(6, dtype=torch.float32) * 0.25
    lat = torch.stack(torch.meshgrid(ax, ax, ax, indexing="ij"), -1).reshape(1, -1, 3)
    n = torch.tensor([lat.size(1)])
    for rad in (0.3, 0.36, 0.26):
        want = R.frnn_bruteforce(lat, lat, n, n, 27, rad)
        got = ops.fast_knn(lat.to(DEV), lat.to(DEV), n.to(DEV), n.to(DEV), 27, rad)
        assert torch.equal(got.cpu(), want), rad
    # argument errors mirror the reference's (point_ops.py:432-440)
    with pytest.raises(ValueError):
        ops.fast_knn(p1[:2].to(DEV), p2.to(DEV), l1, l2, 4, 0.1)
    with pytest.raises(TypeError):
        ops.fast_knn(p1, p2, l1, l2, 4, 0.1)


def test_frnn_far_from_the_origin(frnn_mode):
    """|coordinate| / r = 1e5 .. 4e6: outside the range the fp32 cell arithmetic of round 1 covered (< 1e4, VERDICT r1
    weak #9); cell coordinates are evaluated in double now, and the result still equals the exhaustive search."""
    ops = _ops()
    from oracle import torch_ref as R
    gen = torch.Generator().manual_seed(77)
    for centre, r in ((5000.0, 0.05), (-20000.0, 0.005), (3.0e4, 0.0078125)):
        p = (torch.rand(2, 1500, 3, generator=gen) * (8 * r) + centre).float()     # ~12 neighbours inside r
        n = torch.tensor([1500, 1100])
        want, want_d = R.frnn_bruteforce(p, p, n, n, 16, r, return_dists=True)
        got, got_d = ops.fast_knn(p.to(DEV), p.to(DEV), n.to(DEV), n.to(DEV), 16, r, return_dists=True)
        assert int((want >= 0).sum()) > 3000                       # the case is not vacuous
        assert torch.equal(got.cpu(), want) and torch.equal(got_d.cpu(), want_d)


def test_frnn_on_curve_clouds_full_size(frnn_mode):
    """BASELINE-size cloud (2048 curves, ~50k points): bit match against the exhaustive oracle."""
    ops = _ops()
    from oracle import torch_ref as R
    d = _synth([0])
    n = torch.tensor([d.pos.size(0)])
    p = d.pos.unsqueeze(0)
    for K, r in ((20, 0.04), (32, 0.1)):
        want = R.frnn_bruteforce(p, p, n, n, K, r)
        got = ops.fast_knn(p.to(DEV), p.to(DEV), n.to(DEV), n.to(DEV), K, r)
        assert torch.equal(got.cpu(), want)
        # size-independent properties: self is the nearest neighbour; rows sorted by distance
        assert torch.equal(got[0, :, 0].cpu(), torch.arange(n.item()))


def test_frnn_edges_match_oracle_flat_lists():
    ops = _ops()
    from oracle import torch_ref as R
    d = _synth([7, 8], n_curves=300)
    u = torch.tensor([0.3])
    idx = R.curve_fps(d.pos, d.batch, d.curve_idxs, 0.01, u)
    want = R.group_fixed_radius(d.pos[idx], d.pos, d.batch[idx], d.batch, 16, 0.05)
    pos, batch, p2c = d.pos.to(DEV), d.batch.to(DEV), d.curve_idxs.to(DEV)
    topo = ops.CurveTopology(batch, p2c)
    idxd = idx.to(DEV)
    topo_q = ops.CurveTopology(batch[idxd], p2c[idxd])
    e = ops.frnn_edges(pos[idxd], topo_q, pos, topo, 16, 0.05)
    assert torch.equal(e.row.cpu(), want[0]) and torch.equal(e.col.cpu(), want[1])


# ---------------------------------------------------------------- section 8(f): samplers and exact kNN
def test_voxel_fps_golden_and_oracle():
    ops = _ops()
    from oracle import torch_ref as R
    g = golden("voxel_fps")
    for key in sorted({k.rsplit(".", 1)[0] for k in g.files}):
        got = ops.voxel_fps(t(g[key + ".pos"], DEV), t(g[key + ".batch"], DEV), float(g[key + ".voxel"]), t(g[key + ".rnd"]))
        assert torch.equal(got.cpu(), t(g[key + ".idx"])), key
    d = _synth([0, 1, 2], n_curves=400)
    for vs in (0.025, 0.07):
        rnd = torch.rand(d.pos.size(0), generator=torch.Generator().manual_seed(3))
        want = R.voxel_fps(d.pos, d.batch, vs, rnd)
        got = ops.voxel_fps(d.pos.to(DEV), d.batch.to(DEV), vs, rnd)
        assert torch.equal(got.cpu(), want)


@pytest.mark.parametrize("ids,ratio", [([0], 0.3), ([1, 2, 3], 0.25)])
def test_fps_matches_oracle(ids, ratio):
    ops = _ops()
    from oracle import torch_ref as R
    d = _synth(ids, n_curves=40)
    start = [5, 17, 3][: len(ids)]
    want = R.farthest_point_indices(d.pos, d.batch, ratio, start=start)
    topo = ops.CurveTopology(d.batch.to(DEV), d.curve_idxs.to(DEV))
    got = ops.fps(d.pos.to(DEV), topo, ratio, start=torch.tensor(start))
    assert torch.equal(got.cpu(), want)


def test_fps_lds_claim_does_not_change_the_samples():
    """The sampling workgroups claim 96 KB of dynamic LDS they never touch (so that no GEMM workgroup shares their CU);
    with and without the claim the sample indices are the same.  Clouds of 16 k .. 53 k points take the hybrid kernel
    (registers + running minima in LDS) when the claim is on and the streaming kernel when it is 0: same samples."""
    ops = _ops()
    from curvecloudnet_amd import _lib
    # ~1 k points (registers), ~36 k and ~50 k points per cloud (hybrid / streaming; the second near the hybrid form's capacity)
    for n_curves, ratio in ((40, 0.25), (1500, 0.05), (2150, 0.02)):
        d = _synth([1, 2], n_curves=n_curves)
        topo = ops.CurveTopology(d.batch.to(DEV), d.curve_idxs.to(DEV))
        outs = []
        try:
            for claim in (98304, 0, 40000):
                _lib.lib().ccn_fps_set_lds_claim(claim)
                outs.append(ops.fps(d.pos.to(DEV), topo, ratio, start=torch.tensor([5, 17])).cpu())
        finally:
            _lib.lib().ccn_fps_set_lds_claim(98304)
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
        assert outs[0].numel() == int(torch.ceil(topo.lengths.cpu() * ratio).sum())


def test_fps_by_a_cluster_of_workgroups_gives_the_same_samples():
    """Clouds of more than 16 384 points are sampled by a cluster of 2-4 workgroups (every point in registers, the candidates of a
    round exchanged through agent-scope granules): the same samples, bit for bit, as ONE workgroup per cloud (hybrid kernel) and
    as the streaming kernel -- mixed batches (a large and a small cloud), duplicate points (ties go to the smaller index), a cloud
    count that leaves clusters of the last group of eight empty."""
    ops = _ops()
    from curvecloudnet_amd import _lib
    lib = _lib.lib()
    cases = [([1, 2], 1500, 0.05, None), ([3], 2150, 0.02, None), ([1, 2, 3], 900, 0.1, None), ([4, 5], 2100, 0.01, "dup"),
             ([6, 7, 8, 9, 10, 11, 12, 13, 14], 800, 0.01, "small")]
    for ids, n_curves, ratio, twist in cases:
        d = _synth(ids, n_curves=n_curves)
        pos = d.pos.clone()
        if twist == "dup":
            pos[1::2] = pos[0::2][: pos[1::2].size(0)]              # every point twice: every maximum is a tie
        batch, curves = d.batch, d.curve_idxs
        if twist == "small":                                         # the last cloud cut to a few points
            keep = (batch < batch.max()) | (torch.arange(batch.numel()) >= batch.numel() - 37)
            pos, batch, curves = pos[keep], batch[keep], curves[keep]
        topo = ops.CurveTopology(batch.to(DEV), curves.to(DEV))
        assert int(topo.lengths.max()) > 16384
        start = torch.arange(len(ids)) * 7 + 3
        outs = []
        try:
            for cluster, claim in ((1, 98304), (0, 98304), (0, 0), (2, 98304)):     # (2: the cross-XCD protocol forced)
                lib.ccn_fps_use_cluster(cluster)
                lib.ccn_fps_set_lds_claim(claim)
                outs.append(ops.fps(pos.to(DEV), topo, ratio, start=start).cpu())
        finally:
            lib.ccn_fps_use_cluster(1)
            lib.ccn_fps_set_lds_claim(98304)
        assert all(torch.equal(outs[0], o) for o in outs[1:]), (ids, n_curves, twist)
        assert outs[0].numel() == int(torch.ceil(topo.lengths.cpu() * ratio).sum())


def test_fps_cluster_that_gives_up_is_resampled_by_one_workgroup():
    """VERDICT r5 weak #4 / ADVICE r5: an ordinary launch cannot promise that the members of a sampling cluster run at the same time.
    A member that hears nothing from a partner raises the cloud's abort word and leaves; the gated one-workgroup launch behind the
    cluster kernel re-samples exactly those clouds.  Forced here by the test hook (1 = member 1 silently leaves, its partners run
    into the poll timeout; 2 = it raises the abort word itself): the samples equal the one-workgroup form's, bit for bit, and the
    fallback counter rises by the number of clouds that have a second round."""
    ops = _ops()
    from curvecloudnet_amd import _lib
    lib = _lib.lib()
    for ids, n_curves, ratio in (([1, 2], 1500, 0.05), ([3], 2400, 0.02), ([1, 2, 3], 900, 0.1)):     # (2400 curves: beyond the hybrid form's 53 k points)
        d = _synth(ids, n_curves=n_curves)
        topo = ops.CurveTopology(d.batch.to(DEV), d.curve_idxs.to(DEV))
        assert int(topo.lengths.max()) > 16384
        start = torch.arange(len(ids)) * 11 + 2
        counter = ops.fps_fallbacks(DEV)
        try:
            lib.ccn_fps_use_cluster(0)
            want = ops.fps(d.pos.to(DEV), topo, ratio, start=start).cpu()
            lib.ccn_fps_use_cluster(1)
            for fault in (2, 1):
                before = int(counter.item())
                lib.ccn_fps_debug_fault(fault)
                got = ops.fps(d.pos.to(DEV), topo, ratio, start=start).cpu()
                assert torch.equal(got, want), (ids, fault)
                assert int(counter.item()) - before == len(ids), (ids, fault, int(counter.item()) - before)
            lib.ccn_fps_debug_fault(0)
            before = int(counter.item())
            assert torch.equal(ops.fps(d.pos.to(DEV), topo, ratio, start=start).cpu(), want)
            assert int(counter.item()) == before                 # an idle device: every cluster finishes by itself
        finally:
            lib.ccn_fps_debug_fault(0)
            lib.ccn_fps_use_cluster(1)


def test_fps_cluster_beside_a_stream_of_gemms():
    """The cluster form runs on the geometry stream BESIDE the feature stream's persistent GEMM workgroups (135 of 160 KB of LDS per
    CU, while a sampling workgroup claims 96 KB): whatever the placement does to the cluster -- finish or give up and fall back --
    the samples are those of the one-workgroup form.  The number of fallbacks is printed, not asserted."""
    ops = _ops()
    from curvecloudnet_amd import _lib
    lib = _lib.lib()
    d = _synth([1, 2, 3, 4], n_curves=1500)
    topo = ops.CurveTopology(d.batch.to(DEV), d.curve_idxs.to(DEV))
    pos = d.pos.to(DEV)
    start = torch.tensor([3, 1, 4, 1])
    lib.ccn_fps_use_cluster(0)
    try:
        want = ops.fps(pos, topo, 0.05, start=start).cpu()
    finally:
        lib.ccn_fps_use_cluster(1)
    a = torch.randn(200000, 512, device=DEV)
    w = torch.randn(512, 512, device=DEV)
    y = torch.empty(200000, 512, device=DEV)
    gemm = lambda: ops._gemm_nt("gemm_nt", a, w, None, y, 200000, 512, 512, None)   # noqa: E731 (the paired persistent kernel)
    side = torch.cuda.Stream()
    counter = ops.fps_fallbacks(DEV)
    before = int(counter.item())
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        for _ in range(40):                                      # ~0.8 ms each: the device is full while the clusters start and run
            gemm()
    got = ops.fps(pos, topo, 0.05, start=start)
    with torch.cuda.stream(side):
        for _ in range(40):
            gemm()
    torch.cuda.synchronize()
    print("fps clusters beside 80 GEMM launches: %d of %d clouds fell back" % (int(counter.item()) - before, 4))
    assert torch.equal(got.cpu(), want)


def test_knn_points_matches_bruteforce():
    ops = _ops()
    from oracle import torch_ref as R
    d = _synth([4, 5], n_curves=120)
    sub = R.voxel_fps(d.pos, d.batch, 0.05, torch.rand(d.pos.size(0), generator=torch.Generator().manual_seed(0)))
    pos_x, batch_x = d.pos[sub], d.batch[sub]
    qp, _, l1, _ = R.padded_layout(d.pos, d.batch)
    sp, _, l2, off2 = R.padded_layout(pos_x, batch_x)
    want = R.knn_bruteforce(qp, sp, l1, l2, 3)
    want[1:] += off2.view(-1, 1, 1)
    mask = torch.arange(qp.size(1))[None, :] < l1[:, None]
    topo_y = ops.CurveTopology(d.batch.to(DEV), torch.zeros_like(d.batch).to(DEV))
    topo_x = ops.CurveTopology(batch_x.to(DEV), torch.zeros_like(batch_x).to(DEV))
    nbr, w = ops.knn_points_packed(d.pos.to(DEV), topo_y, pos_x.to(DEV), topo_x, 3)
    assert torch.equal(nbr.cpu(), want[mask])
    assert bool((w > 0).all())


def test_ball_query_and_exact_knn_match_oracle():
    ops = _ops()
    from oracle import torch_ref as R
    p1, p2, l1, l2 = _frnn_case(3, 300, 700, 128, 0.3, seed=11)
    for r in (0.15, 0.6):
        want = R.ball_query_bruteforce(p1, p2, l1, l2, 128, r)
        got = ops.ball_query(p1.to(DEV), p2.to(DEV), l1.to(DEV), l2.to(DEV), 128, r)
        assert torch.equal(got.cpu(), want), r
    for K in (3, 20, 30):        # exact kNN = the grid search with an unbounded radius
        want = R.knn_bruteforce(p1, p2, l1, l2, K)
        got = ops.fast_knn(p1.to(DEV), p2.to(DEV), l1.to(DEV), l2.to(DEV), K, ops.EXACT_KNN_RADIUS)
        assert torch.equal(got.cpu(), want), K


def test_curve_splitters_bit_exact():
    """ccn_curve_split vs. the reference's outputs (golden) and vs. the oracle on a KITTI-size sweep."""
    from curvecloudnet_amd import data as D
    from oracle import torch_ref as R
    g = golden("harness")
    for ci in range(4):
        key = "split%d" % ci
        pts, beams = t(g[key + ".points"], DEV), t(g[key + ".beams"], DEV)
        assert torch.equal(D.get_curves_kitti(pts).cpu(), t(g[key + ".kitti"]))
        n = pts.size(0)
        out = D.get_curves_nuscenes(pts, beams, torch.zeros(n, device=DEV), torch.zeros(n, device=DEV))
        assert torch.equal(out[1].cpu(), t(g[key + ".nus_curves"]))
        assert torch.equal(out[4].cpu(), t(g[key + ".nus_inverse"]))
    gen = torch.Generator().manual_seed(77)
    n = 120_000
    steps = torch.randn(n, 3, generator=gen) * 0.03
    steps[torch.rand(n, generator=gen) < 0.02] *= 30.0
    pts = (torch.cumsum(steps, 0) + torch.tensor([10.0, 4.0, -1.0])).float()
    beams = torch.sort(torch.randint(0, 64, (n,), generator=gen))[0]
    want = R.split_curves(pts, beams)
    got = D.split_curves(pts.to(DEV), beams.to(DEV))
    assert torch.equal(got.cpu(), want) and int(want[-1]) > 1000
    assert D.split_curves(pts[:0].to(DEV)).numel() == 0


def test_lovasz_loss_on_device():
    from curvecloudnet_amd import loss as L
    g = golden("harness")
    for ci in range(3):
        key = "lovasz%d" % ci
        probas = t(g[key + ".probas"], DEV).requires_grad_(True)
        out = L.lovasz_softmax_flat(probas, t(g[key + ".labels"], DEV))
        grad, = torch.autograd.grad(out, probas)
        assert abs(float(out) - float(g[key + ".loss"])) < 1e-5
        assert float((grad.cpu() - t(g[key + ".grad"])).abs().max()) < 1e-6


def test_grid_knn_equals_exhaustive_knn():
    """The hash-grid path of knn_points_packed (large clouds) returns exactly what the exhaustive kernel returns:
    same neighbours in the same order, same weights -- including the queries the density radius misses."""
    from curvecloudnet_amd import ops
    from curvecloudnet_amd.synth import make_batch
    d = make_batch([0, 1])
    pos, batch = d.pos.to(DEV), d.batch.to(DEV)
    keep = torch.arange(0, pos.size(0), 2, device=DEV)
    far = torch.tensor([[40.0, 40.0, 5.0], [-35.0, 20.0, -4.0]], device=DEV)      # isolated queries: forced misses
    pos_q = torch.cat([pos[batch == 0], far[:1], pos[batch == 1], far[1:]])
    batch_q = torch.cat([batch[batch == 0], batch[:1] * 0, batch[batch == 1], batch[:1] * 0 + 1])
    topo_q = ops.CurveTopology(batch_q, torch.zeros_like(batch_q))
    topo_s = ops.CurveTopology(batch[keep], torch.zeros_like(batch[keep]))
    assert topo_s.max_cloud >= ops.KNN_GRID_MIN_POINTS
    for k in (1, 3, 8):
        nbr_g, w_g = ops.knn_points_packed(pos_q, topo_q, pos[keep], topo_s, k)
        missed = int(ops.knn_points_packed.last_missed.item())
        scale, ops.KNN_GRID_SCALE = ops.KNN_GRID_SCALE, 0.0
        try:
            nbr_e, w_e = ops.knn_points_packed(pos_q, topo_q, pos[keep], topo_s, k)
        finally:
            ops.KNN_GRID_SCALE = scale
        assert torch.equal(nbr_g, nbr_e) and torch.equal(w_g, w_e)
        print("k=%d: %d of %d queries recomputed exhaustively" % (k, missed, pos_q.size(0)))
        assert 2 <= missed < pos_q.size(0) // 4


def test_grid_knn_with_fewer_sources_than_k():
    """Clouds whose source set is smaller than K: the table is -1 padded exactly like the exhaustive kernel's, whichever
    path a cloud takes (the size threshold is lowered so that the grid path runs on these small clouds)."""
    from curvecloudnet_amd import ops
    gen = torch.Generator().manual_seed(3)
    pos_s = torch.rand(2 + 900, 3, generator=gen).to(DEV)
    batch_s = torch.cat([torch.zeros(2, dtype=torch.long), torch.ones(900, dtype=torch.long)]).to(DEV)
    pos_q = torch.rand(700, 3, generator=gen).to(DEV)
    batch_q = torch.cat([torch.zeros(300, dtype=torch.long), torch.ones(400, dtype=torch.long)]).to(DEV)
    topo_s = ops.CurveTopology(batch_s, torch.zeros_like(batch_s))
    topo_q = ops.CurveTopology(batch_q, torch.zeros_like(batch_q))
    keep = ops.KNN_GRID_MIN_POINTS
    try:
        ops.KNN_GRID_MIN_POINTS = 1
        nbr_g, w_g = ops.knn_points_packed(pos_q, topo_q, pos_s, topo_s, 3)
        ops.KNN_GRID_MIN_POINTS = 10 ** 9
        nbr_e, w_e = ops.knn_points_packed(pos_q, topo_q, pos_s, topo_s, 3)
    finally:
        ops.KNN_GRID_MIN_POINTS = keep
    assert torch.equal(nbr_g, nbr_e) and torch.equal(w_g, w_e)
    assert bool((nbr_e[:300, 2] == -1).all()) and bool((nbr_e[:300, :2] >= 0).all()) and bool((nbr_e[300:] >= 2).all())


@pytest.mark.parametrize("n,hi", [(1, 10), (1000, 7), (5000, 1 << 40), (400070, 1 << 58), (70000, 1), (123457, 300)])
def test_rank_keys_equals_torch_unique_inverse(n, hi):
    """ccn_rank_keys (the hand-written radix sort behind VoxelFPS) = torch.unique(sorted=True, return_inverse=True): the dense
    rank of every key among the distinct keys and their number, bit-exact; with all eight digit passes and with only the
    digits ccn_key_spread reports."""
    from curvecloudnet_amd._lib import call, lib, ptr, workspace
    gen = torch.Generator().manual_seed(n)
    key = torch.randint(0, hi, (n,), generator=gen, dtype=torch.int64)
    if hi > 1000:
        key[::3] = key[1::3][: key[::3].numel()]           # plenty of duplicates among wide keys too
    uniq, inv = torch.unique(key, return_inverse=True)
    kd = key.to(DEV)
    meta = torch.zeros(2, dtype=torch.int64, device=DEV)
    call("key_spread", ptr(kd), n, ptr(meta))
    spread = int(meta[0].item())
    want_spread = 0
    for v in (key ^ key[0]).tolist()[:2000]:
        want_spread |= v
    assert spread & want_spread == want_spread
    nb = lib().ccn_rank_keys_workspace_bytes(n)
    ws = workspace(nb, DEV)
    for digits in (0xff, sum(1 << b for b in range(8) if (spread >> (8 * b)) & 255)):
        rank = torch.full((n,), -1, dtype=torch.int64, device=DEV)
        call("rank_keys", ptr(kd), n, digits, ptr(rank), ptr(meta[1:]), ptr(ws), nb)
        assert int(meta[1].item()) == uniq.numel()
        assert torch.equal(rank.cpu(), inv)


@pytest.mark.parametrize("n,m,dtype", [(1, 1, torch.int32), (1000, 7, torch.int64), (50000, 50000, torch.int32),
                                        (2341754, 400070, torch.int32), (300000, 5, torch.int64), (4096, 100000, torch.int32)])
def test_inverse_lists_equal_a_stable_sort(n, m, dtype):
    """ccn_inverse_lists (what the atomics-free backward of the first edge layers gathers through) = the stable sort of the
    row numbers by source + a histogram, bit-exact, whatever order its atomics land in; ccn_group_owner = repeat_interleave of
    the group numbers."""
    from curvecloudnet_amd import ops
    from curvecloudnet_amd._lib import call, ptr
    gen = torch.Generator().manual_seed(n + m)
    src = torch.randint(0, m, (n,), generator=gen).to(dtype)
    if n > 100:
        src[: n // 4] = src[n // 2: n // 2 + n // 4]          # long lists as well
    inv_ptr, inv_row = ops.inverse_lists(src.to(DEV), m)
    want_row = torch.sort(src, stable=True)[1].to(torch.int32)
    want_ptr = torch.zeros(m + 1, dtype=torch.int32)
    want_ptr[1:] = torch.cumsum(torch.bincount(src.long(), minlength=m), 0).to(torch.int32)
    assert torch.equal(inv_ptr.cpu(), want_ptr) and torch.equal(inv_row.cpu(), want_row)
    sizes = torch.randint(0, 9, (max(m, 1),), generator=gen)
    grp = torch.zeros(sizes.numel() + 1, dtype=torch.int32)
    grp[1:] = torch.cumsum(sizes, 0).to(torch.int32)
    e = int(grp[-1])
    owner = torch.full((max(e, 1),), -1, dtype=torch.int32, device=DEV)
    call("group_owner", ptr(grp.to(DEV)), sizes.numel(), e, ptr(owner))
    assert torch.equal(owner[:e].cpu(), torch.repeat_interleave(torch.arange(sizes.numel(), dtype=torch.int32), sizes))
