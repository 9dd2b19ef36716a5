"""Round 5: the tail split of the paired fp32 kernel (``ccn_gemm_nt_ws`` and its ``_acc`` / ``_xf`` / ``_red`` siblings;
csrc/ccn_gemm.hip ``launch_glds_pair``).  Same products as the entries without scratch (F.linear inside PyG MLP,
reference src/models/base.py:90-125); the last, partly filled round of 128 x 128 tiles is cut along K.

Checked here: (i) the split really happens for the shapes below (``ccn_gemm_nt_split_parts``); (ii) every variant agrees with
the unsplit entry to fp32 re-association (the parts are summed in part order: a different grouping of the same K chain) and
with an fp64 product within the bound of tests/test_gpu_gemm_f64.py; (iii) two runs give the SAME bits (the last part to
arrive differs from run to run, the summation order must not); (iv) the scratch's counters are left at zero; (v) BatchNorm
statistics / fused backward sums come out of the finishing part's epilogue like any other tile's."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

# (M, N, K, parts expected): 664 tiles = one full round + 152 -> 3 parts; 200 tiles, no full round -> 2 parts;
# (K >= 512 only: below that the fix-up costs what the split saves); K % 32 != 0 (register-staged remainder slice in the last part); 1100 tiles = 2 rounds + 76 -> 4 parts (K = 512: 16 slices)
SHAPES = [(10550, 1024, 1024, 3), (2560, 1280, 512, 2), (10550, 1024, 1027, 3), (35151, 512, 512, 4)]


def _operands(rows, cols, gen, scale=1.0):
    from curvecloudnet_amd.ops import _rows
    t = _rows(rows, cols, DEV, zero=True)
    t[:, :cols].copy_((torch.randn(rows, cols, generator=gen, device=DEV) *
                       torch.pow(10.0, torch.rand(rows, cols, generator=gen, device=DEV) * 3 - 2)) * scale)
    return t


def _scratch():
    from curvecloudnet_amd._lib import lib
    n = int(lib().ccn_gemm_nt_split_workspace_bytes())
    return torch.zeros(n, dtype=torch.uint8, device=DEV), n


def _rel_err(y, x, w, M, N, K, bias=None):
    worst = 0.0
    step = max(1, (1 << 27) // (N * 8))
    wd = w[:, :K].double()
    for r0 in range(0, M, step):
        xd = x[r0:r0 + step, :K].double()
        ref = xd @ wd.t()
        scale = (xd.abs() @ wd.abs().t()).clamp_min(1e-300)
        if bias is not None:
            ref = ref + bias.double()[None, :]
            scale = scale + bias.double().abs()[None, :]
        worst = max(worst, float(((y[r0:r0 + step, :N].double() - ref).abs() / scale).max()))
    return worst


@pytest.mark.parametrize("M,N,K,parts", SHAPES)
def test_split_tail_product_statistics_and_determinism(M, N, K, parts):
    from curvecloudnet_amd._lib import call, lib, ptr
    from curvecloudnet_amd.ops import _ld, _rows, _stats_buffer
    ws, nb = _scratch()
    assert lib().ccn_gemm_nt_split_parts(M, N, K, nb) == parts
    assert lib().ccn_gemm_nt_split_parts(M, N, K, 0) == 1
    gen = torch.Generator(device=DEV).manual_seed(M + 7 * N + K)
    x, w = _operands(M, K, gen), _operands(N, K, gen, K ** -0.5)
    bias = torch.randn(N, generator=gen, device=DEV)
    y0, y1, y2 = (_rows(M, N, DEV) for _ in range(3))
    for t in (y0, y1, y2):
        t.fill_(float("nan"))
    st0, st1 = _stats_buffer(M, N, DEV), _stats_buffer(M, N, DEV)
    call("gemm_nt", ptr(x), _ld(x), ptr(w), _ld(w), ptr(bias), ptr(y0), _ld(y0), M, N, K, ptr(st0))
    call("gemm_nt_ws", ptr(x), _ld(x), ptr(w), _ld(w), ptr(bias), ptr(y1), _ld(y1), M, N, K, ptr(st1), ptr(ws), nb)
    call("gemm_nt_ws", ptr(x), _ld(x), ptr(w), _ld(w), ptr(bias), ptr(y2), _ld(y2), M, N, K, None, ptr(ws), nb)
    torch.cuda.synchronize()
    assert not torch.isnan(y1[:, :N]).any()
    assert torch.equal(y1[:, :N], y2[:, :N]), "two runs of the split product differ in bits"
    assert int(ws[:4096].view(torch.int32).abs().sum()) == 0, "the scratch's counters are not left at zero"
    err = _rel_err(y1, x, w, M, N, K, bias)
    print("gemm_nt_ws %dx%dx%d (%d parts): max |err| / sum|a||w| = %.3g, max |split - unsplit| = %.3g"
          % (M, N, K, parts, err, float((y1[:, :N] - y0[:, :N]).abs().max())))
    assert err < 2.5e-6
    # the tail tiles take another grouping of the K chain; everything else is the same code path, so most elements are equal
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    same = float((y1[:, :N] == y0[:, :N]).float().mean())
    assert same >= 0.9 * (tiles // 512 * 512) / tiles, same
    assert float((y1[:, :N] - y0[:, :N]).abs().max()) < 1e-4 * float(y0[:, :N].abs().max())
    # BatchNorm partial statistics: one row per 128-row block, written by whichever part finished the tile
    nparts = int(lib().ccn_stats_rows(M))
    s0 = st0[:nparts * 2 * N].view(nparts, 2 * N).sum(0)
    s1 = st1[:nparts * 2 * N].view(nparts, 2 * N).sum(0)
    ref = torch.cat([y1[:, :N].double().sum(0), (y1[:, :N].double() ** 2).sum(0)])
    yard = torch.cat([y1[:, :N].double().abs().sum(0), (y1[:, :N].double() ** 2).sum(0)]).clamp_min(1.0)   # (fp32 sums per 32 rows)
    assert float(((s1 - ref).abs() / yard).max()) < 1e-5
    assert float(((s1 - s0).abs() / yard).max()) < 1e-5


def test_split_tail_in_the_transforming_accumulating_and_reducing_variants():
    """``_xf`` (deferred BatchNorm + activation on the A fragments), ``_acc`` (Y += ...) and ``_red`` (the previous layer's
    BatchNorm-backward sums in the epilogue) through the split, each against its unsplit entry."""
    from curvecloudnet_amd._lib import call, lib, ptr
    from curvecloudnet_amd.ops import ACT, LEAKY_SLOPE, _ld, _rows, _stats_buffer
    M, N, K = 10550, 1024, 512
    ws, nb = _scratch()
    assert lib().ccn_gemm_nt_split_parts(M, N, K, nb) >= 2
    assert lib().ccn_gemm_nt_xf_ok(K, K, M, N, K)
    gen = torch.Generator(device=DEV).manual_seed(5)
    x, w = _operands(M, K, gen), _operands(N, K, gen, K ** -0.5)
    act = ACT["relu"]
    # ---- xf: z = act(x * scale + shift) applied between LDS and the matrix cores
    sc = torch.rand(K, generator=gen, device=DEV) + 0.5
    sh = torch.randn(K, generator=gen, device=DEV) * 0.1
    ya, yb = _rows(M, N, DEV), _rows(M, N, DEV)
    call("gemm_nt_xf", ptr(x), _ld(x), ptr(sc), ptr(sh), act, LEAKY_SLOPE, ptr(w), _ld(w), None, ptr(ya), _ld(ya), M, N, K, None)
    call("gemm_nt_xf_ws", ptr(x), _ld(x), ptr(sc), ptr(sh), act, LEAKY_SLOPE, ptr(w), _ld(w), None, ptr(yb), _ld(yb), M, N, K,
         None, ptr(ws), nb)
    z = _rows(M, K, DEV)
    call("bn_act_fwd", ptr(x), _ld(x), M, K, ptr(sc), ptr(sh), act, LEAKY_SLOPE, ptr(z), _ld(z))
    yc = _rows(M, N, DEV)
    call("gemm_nt_ws", ptr(z), _ld(z), ptr(w), _ld(w), None, ptr(yc), _ld(yc), M, N, K, None, ptr(ws), nb)
    torch.cuda.synchronize()
    assert torch.equal(yb[:, :N], yc[:, :N]), "deferred activation + split differs from written activation + split"
    assert _rel_err(yb, z, w, M, N, K) < 2.5e-6
    assert float((ya[:, :N] - yb[:, :N]).abs().max()) < 1e-3 * float(ya[:, :N].abs().max())
    # ---- acc: Y += A W^T
    base = torch.randn(M, N, generator=gen, device=DEV)
    yd = _rows(M, N, DEV)
    yd[:, :N].copy_(base)
    call("gemm_nt_acc_ws", ptr(z), _ld(z), ptr(w), _ld(w), ptr(yd), _ld(yd), M, N, K, ptr(ws), nb)
    torch.cuda.synchronize()
    tol = 1e-5 * float(yc[:, :N].abs().max() + base.abs().max())
    assert float((yd[:, :N] - (base + yc[:, :N])).abs().max()) < tol
    # ---- red: product + the previous layer's BatchNorm-backward sums
    Kr = 256                                      # output width of the data-gradient product = that layer's width
    dy, wt = _operands(M, N, gen), _operands(Kr, N, gen, N ** -0.5)
    yprev = _operands(M, Kr, gen)
    par = torch.stack([torch.rand(Kr, generator=gen, device=DEV) + 0.5, torch.randn(Kr, generator=gen, device=DEV) * 0.1,
                       torch.randn(Kr, generator=gen, device=DEV) * 0.1, torch.rand(Kr, generator=gen, device=DEV) + 0.5]).contiguous()
    M2 = 39000                                                    # 305 x 2 = 610 tiles -> 98 in the tail round
    dy2, yprev2 = _operands(M2, N, gen), _operands(M2, Kr, gen)
    assert lib().ccn_gemm_nt_split_parts(M2, Kr, N, nb) >= 2
    dx0, dx1 = _rows(M2, Kr, DEV), _rows(M2, Kr, DEV)
    s0, s1 = _stats_buffer(M2, Kr, DEV), _stats_buffer(M2, Kr, DEV)
    call("gemm_nt_red", ptr(dy2), _ld(dy2), ptr(wt), _ld(wt), ptr(dx0), _ld(dx0), M2, Kr, N, ptr(yprev2), _ld(yprev2), ptr(par),
         act, LEAKY_SLOPE, ptr(s0))
    call("gemm_nt_red_ws", ptr(dy2), _ld(dy2), ptr(wt), _ld(wt), ptr(dx1), _ld(dx1), M2, Kr, N, ptr(yprev2), _ld(yprev2), ptr(par),
         act, LEAKY_SLOPE, ptr(s1), ptr(ws), nb)
    torch.cuda.synchronize()
    assert _rel_err(dx1, dy2, wt, M2, Kr, N) < 2.5e-6
    yard = torch.cat([dx1[:, :Kr].double().abs().sum(0)] * 2).clamp_min(1.0)
    assert float(((s1[:2 * Kr] - s0[:2 * Kr]).abs() / yard).max()) < 1e-4
    assert int(ws[:4096].view(torch.int32).abs().sum()) == 0
    del dy, yprev
