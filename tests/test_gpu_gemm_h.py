"""The 16-bit STORAGE kernels (csrc/ccn_gemm_h.hip) through the C-ABI with raw pointers: casts, ccn_gemm_nt_h (fp32 and
16-bit results, statistics, K remainders), ccn_gemm_tn_h (transposed LDS reads), the 16-bit BatchNorm passes.

Operand lane maps are checked with EXACT small-integer data that is asymmetric in every index (a swapped row / column /
contraction map cannot pass), then random data is held against fp64 products of the rounded operands."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _api():
    from curvecloudnet_amd._lib import call, lib, ptr, workspace
    return call, lib, ptr, workspace


def _rows16(rows, cols, dtype=torch.bfloat16):
    ld = (cols + 7) // 8 * 8
    buf = torch.zeros((rows, ld), dtype=dtype, device=DEV)
    return buf, ld


def _to16(x, dtype=torch.bfloat16):
    """fp32 (rows, cols) on the CPU -> zero-padded 16-bit rows on the GPU via ccn_cast_rows_h, checked against torch's cast."""
    call, lib, ptr, _ = _api()
    rows, cols = x.shape
    xd = x.to(DEV).contiguous()
    buf, ld = _rows16(rows, cols, dtype)
    buf.fill_(7.0)                                            # (the cast must overwrite the padding columns with zeros)
    call("cast_rows_h", ptr(xd), cols, rows, cols, ptr(buf), ld, 1 if dtype == torch.float16 else 0)
    assert torch.equal(buf[:, :cols].cpu(), x.to(dtype)), "cast_rows_h differs from torch's round-to-nearest-even"
    assert bool((buf[:, cols:] == 0).all())
    return buf, ld


@pytest.mark.parametrize("rows,cols", [(5, 3), (1000, 64), (777, 259), (33, 8)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_cast_rows(rows, cols, dtype):
    gen = torch.Generator().manual_seed(rows + cols)
    _to16(torch.randn(rows, cols, generator=gen) * 3, dtype)


def _int_matrix(rows, cols, seed, lo=-3, hi=4):
    gen = torch.Generator().manual_seed(seed)
    m = torch.randint(lo, hi, (rows, cols), generator=gen).float()
    # asymmetric in both indices: a distinct ramp per row and per column
    return m + (torch.arange(rows)[:, None] % 5).float() - (torch.arange(cols)[None, :] % 3).float()


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 192, 128), (1000, 259, 72), (129, 64, 200), (5000, 256, 256),
                                   (70000, 128, 320)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_gemm_nt_h_exact_integers_and_statistics(M, N, K, dtype):
    """Small integers are exact in bf16 / fp16 and their sums exact in fp32: the product must EQUAL the integer result, for
    M / N / K that are not multiples of the tile and slice (clamped rows, zero-filled K remainder), with and without
    bias and statistics, as fp32 rows and as 16-bit rows."""
    call, lib, ptr, _ = _api()
    f16 = 1 if dtype == torch.float16 else 0
    a, w = _int_matrix(M, K, 1), _int_matrix(N, K, 2)
    a16, lda = _to16(a, dtype)
    w16, ldw = _to16(w, dtype)
    bias = torch.arange(N, dtype=torch.float32) % 7
    want = (a.double() @ w.double().t())
    assert float(want.abs().max()) < 2 ** 23
    y = torch.full((M, N + 5), -1.0, device=DEV)
    nparts = lib().ccn_stats_rows(M)
    stats = torch.zeros((nparts + 1) * 2 * N, dtype=torch.float64, device=DEV)
    call("gemm_nt_h", ptr(a16), lda, ptr(w16), ldw, ptr(bias.to(DEV)), ptr(y), N + 5, M, N, K, ptr(stats), f16, 0)
    got = y[:, :N].cpu().double()
    assert torch.equal(got, want + bias.double()[None, :]), "fp32 result"
    assert bool((y[:, N:] == -1.0).all()), "columns beyond N were written"
    parts = stats[: nparts * 2 * N].view(nparts, 2, N).sum(0).cpu()
    assert torch.allclose(parts[0], got.sum(0), rtol=1e-6, atol=1e-3)
    assert torch.allclose(parts[1], (got * got).sum(0), rtol=1e-5, atol=1e-2)
    # 16-bit result (the data-gradient form: swapped MFMA operands, 8-byte stores): the rounding of the exact product
    ld16 = (N + 3) // 4 * 4
    y16 = torch.full((M, ld16), -1.0, dtype=dtype, device=DEV)
    call("gemm_nt_h", ptr(a16), lda, ptr(w16), ldw, None, ptr(y16), ld16, M, N, K, None, f16, 1)
    assert torch.equal(y16[:, :N].cpu(), want.float().to(dtype)), "16-bit result"
    assert bool((y16[:, N:] == -1.0).all())


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 192, 128), (1000, 259, 72), (129, 64, 200), (5000, 256, 256),
                                   (70000, 128, 320), (40000, 40, 72)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("act", [1, 2])
def test_batchnorm_layer_forward_without_its_fp32_intermediate(M, N, K, dtype, act):
    """Round 6: ccn_gemm_nt_h_stats = the statistics of ccn_gemm_nt_h with nothing stored (the same fp32 sums: bit-identical
    partial rows); ccn_gemm_nt_h_bnact = ccn_gemm_nt_h followed by ccn_bn_act_fwd / ccn_bn_act_fwd_h, as fp32 rows and as 16-bit
    rows -- the same bits (one fused multiply-add on the same fp32 sum), for shapes that are not multiples of the tile / slice,
    with a bias (folded into the shift by the caller)."""
    call, lib, ptr, _ = _api()
    f16 = 1 if dtype == torch.float16 else 0
    gen = torch.Generator().manual_seed(M + N + K + act)
    a16, lda = _to16(torch.randn(M, K, generator=gen), dtype)
    w16, ldw = _to16(torch.randn(N, K, generator=gen) / K ** 0.5, dtype)
    bias = (torch.randn(N, generator=gen) * 0.2).to(DEV)
    scale, shift = (torch.rand(N, generator=gen) + 0.5).to(DEV), (torch.randn(N, generator=gen) * 0.3).to(DEV)
    nparts = lib().ccn_stats_rows(M)
    for b in (bias, None):
        y = torch.empty(M, N, device=DEV)
        stats = torch.zeros((nparts + 1) * 2 * N, dtype=torch.float64, device=DEV)
        call("gemm_nt_h", ptr(a16), lda, ptr(w16), ldw, ptr(b), ptr(y), N, M, N, K, ptr(stats), f16, 0)
        stats2 = torch.zeros_like(stats)
        call("gemm_nt_h_stats", ptr(a16), lda, ptr(w16), ldw, ptr(b), M, N, K, ptr(stats2), f16)
        assert torch.equal(stats2, stats), "statistics-only pass"
        shift_eff = shift if b is None else torch.addcmul(shift, b, scale)
        # reference: the three-kernel form on the product WITHOUT the bias + the folded shift (the fused kernel's own definition),
        # which equals BatchNorm of (product + bias) up to one fp32 rounding of the folded constant
        y0 = torch.empty(M, N, device=DEV)
        call("gemm_nt_h", ptr(a16), lda, ptr(w16), ldw, None, ptr(y0), N, M, N, K, None, f16, 0)
        z32 = torch.empty(M, N, device=DEV)
        call("bn_act_fwd", ptr(y0), N, M, N, ptr(scale), ptr(shift_eff), act, 0.01, ptr(z32), N)
        got32 = torch.full((M, N + 3), -7.0, device=DEV)
        call("gemm_nt_h_bnact", ptr(a16), lda, ptr(w16), ldw, ptr(scale), ptr(shift_eff), act, 0.01, ptr(got32), N + 3, None, 0, M, N, K, f16, 0)
        assert torch.equal(got32[:, :N], z32), float((got32[:, :N] - z32).abs().max())
        assert bool((got32[:, N:] == -7.0).all())
        z16, ldz = _rows16(M, N, dtype)
        call("bn_act_fwd_h", ptr(y0), N, M, N, ptr(scale), ptr(shift_eff), act, 0.01, ptr(z16), ldz, f16)
        got16, ldg = _rows16(M, N, dtype)
        got16.fill_(3.0)
        call("gemm_nt_h_bnact", ptr(a16), lda, ptr(w16), ldw, ptr(scale), ptr(shift_eff), act, 0.01, ptr(got16), ldg, None, 0, M, N, K, f16, 1)
        assert torch.equal(got16[:, :N], z16[:, :N]), "16-bit rows"
        # ... and with the pre-activation as a second 16-bit result (what a ReLU layer keeps for its backward pass)
        t32 = torch.empty(M, N, device=DEV)
        call("bn_act_fwd", ptr(y0), N, M, N, ptr(scale), ptr(shift_eff), 0, 0.01, ptr(t32), N)
        got16.fill_(3.0)
        t16, ldt = _rows16(M, N, dtype)
        t16.fill_(3.0)
        call("gemm_nt_h_bnact", ptr(a16), lda, ptr(w16), ldw, ptr(scale), ptr(shift_eff), act, 0.01, ptr(got16), ldg, ptr(t16), ldt, M, N, K, f16, 1)
        assert torch.equal(got16[:, :N], z16[:, :N]) and torch.equal(t16[:, :N], t32.to(dtype)), "two 16-bit results"
        if b is not None:      # against BatchNorm of the biased product: the folding is one rounding of a constant away
            zb = torch.empty(M, N, device=DEV)
            call("bn_act_fwd", ptr(y), N, M, N, ptr(scale), ptr(shift), act, 0.01, ptr(zb), N)
            assert float((got32[:, :N] - zb).abs().max()) <= 4e-6 * max(1.0, float(zb.abs().max()))


@pytest.mark.parametrize("rows,C", [(1000, 64), (4099, 259), (300, 8), (70000, 128)])
@pytest.mark.parametrize("act", [1, 2])
def test_batchnorm_backward_from_the_layer_output(rows, C, act):
    """ccn_bn_act_bwd_reduce_hz / _apply_hz take the layer's OUTPUT z in place of its pre-normalisation product y.  With z held in
    fp32 they reproduce the y-passes to fp32 rounding (t = act^-1(z), xhat = (t - beta) / gamma); with z rounded to bf16 / fp16 they
    equal the y-passes evaluated on the y that this rounded z stands for."""
    call, lib, ptr, _ = _api()
    gen = torch.Generator().manual_seed(rows + C + act)
    y = torch.randn(rows, C, generator=gen).to(DEV)
    par = torch.stack([torch.rand(C, generator=gen) + 0.5, torch.randn(C, generator=gen) * 0.3,
                       torch.randn(C, generator=gen) * 0.1, torch.rand(C, generator=gen) + 0.5]).to(DEV)   # scale shift mean rstd
    nparts = lib().ccn_stats_rows(rows)
    g32 = torch.randn(rows, C, generator=gen).to(DEV)
    g16, ldg = _to16(g32.cpu())
    g16f = g16[:, :C].float().contiguous()
    z32 = torch.empty(rows, C, device=DEV)
    call("bn_act_fwd", ptr(y), C, rows, C, ptr(par[0]), ptr(par[1]), act, 0.01, ptr(z32), C)

    def y_passes(yy, gg):
        s = torch.zeros((nparts + 1) * 2 * C, dtype=torch.float64, device=DEV)
        call("bn_act_bwd_reduce", ptr(gg), C, ptr(yy), C, rows, C, ptr(par[0]), ptr(par[1]), ptr(par[2]), ptr(par[3]), act, 0.01, ptr(s))
        dy = torch.empty(rows, C, device=DEV)
        dgb = torch.zeros(2, C, device=DEV)
        call("bn_act_bwd_apply_ex", ptr(gg), C, ptr(yy), C, rows, C, ptr(par[0]), ptr(par[1]), ptr(par[2]), ptr(par[3]), act, 0.01,
             ptr(s), float(rows), 1, 0, ptr(dy), C, ptr(dgb[0]), ptr(dgb[1]))
        return s[: 2 * C].clone(), dy, dgb

    # LeakyReLU (act 2): the output z is invertible -- z rows, z_pre = 0.  ReLU (act 1): xhat of a clipped element is gone from z,
    # and BatchNorm's backward needs it for EVERY row: such a layer hands over its 16-bit PRE-activation t -- t rows, z_pre = 1.
    pre = 1 if act == 1 else 0
    t32 = torch.empty(rows, C, device=DEV)
    call("bn_act_fwd", ptr(y), C, rows, C, ptr(par[0]), ptr(par[1]), 0, 0.01, ptr(t32), C)
    src32 = t32 if pre else z32
    for zt, zdt in ((3, torch.float32), (1, torch.bfloat16), (2, torch.float16)):
        if zt == 3:
            z, ldz = src32, C
            zval = src32
        else:
            z, ldz = _rows16(rows, C, zdt)
            z[:, :C] = src32.to(zdt)
            zval = z[:, :C].float()
        # the y this operand stands for: t = the operand itself (pre-activation) or act^-1(z); y = (t - shift) / scale
        t = zval if pre else torch.where(zval > 0, zval, zval * (1.0 / torch.tensor(0.01, dtype=torch.float32)).item())
        y_of_z = ((t - par[1]) / par[0]).contiguous()
        for dz16, gsrc, ldsrc, gval in ((1, g16, ldg, g16f), (0, g32, C, g32)):
            want_s, want_dy, want_dgb = y_passes(y_of_z, gval)
            s = torch.zeros((nparts + 1) * 2 * C, dtype=torch.float64, device=DEV)
            call("bn_act_bwd_reduce_hz", ptr(gsrc), dz16, ldsrc, ptr(z), zt, pre, ldz, rows, C, ptr(par[0]), ptr(par[1]), ptr(par[2]),
                 ptr(par[3]), act, 0.01, ptr(s))
            scale_s = float(want_s.abs().max()) + rows ** 0.5
            # (an element whose t is within rounding of 0 may take the other slope on one side: its g is one term of rows)
            assert float((s[: 2 * C] - want_s).abs().max()) <= 2e-5 * scale_s + 4.0, (zt, dz16, float((s[: 2 * C] - want_s).abs().max()))
            dy16, lddy = _rows16(rows, C)
            dgb = torch.full((2, C), 9.0, device=DEV)
            call("bn_act_bwd_apply_hz", ptr(gsrc), dz16, ldsrc, ptr(z), zt, pre, ldz, rows, C, ptr(par[0]), ptr(par[1]), ptr(par[2]),
                 ptr(par[3]), act, 0.01, ptr(_sums_like(want_s, nparts, C)), float(rows), 1, 0,
                 ptr(dy16), lddy, ptr(dgb[0]), ptr(dgb[1]))
            diff = (dy16[:, :C].float() - want_dy).abs()
            tol = want_dy.abs() * 2 ** -7 + 2e-5
            # (y_of_z * scale + shift reproduces t only to rounding: where |t| < 1e-6 the reference may sit on the other slope)
            kink = (t.abs() < 1e-6)
            assert bool(((diff <= tol) | kink).all()), (zt, dz16, float(diff.max()))
            assert torch.allclose(dgb, want_dgb, rtol=1e-6, atol=1e-6)
            assert bool((dy16[:, C:] == 0).all())
        if zt == 3:      # fp32 operand: the y-passes on the TRUE y, to fp32 rounding of the inversion
            true_s, true_dy, _ = y_passes(y, g32)
            s = torch.zeros((nparts + 1) * 2 * C, dtype=torch.float64, device=DEV)
            call("bn_act_bwd_reduce_hz", ptr(g32), 0, C, ptr(src32), 3, pre, C, rows, C, ptr(par[0]), ptr(par[1]), ptr(par[2]), ptr(par[3]),
                 act, 0.01, ptr(s))
            assert float((s[: 2 * C] - true_s).abs().max()) <= 1e-4 * (float(true_s.abs().max()) + rows ** 0.5)


def _sums_like(totals, nparts, C):
    """a sums buffer (2 C totals + scratch) holding ``totals``"""
    s = torch.zeros((nparts + 1) * 2 * C, dtype=torch.float64, device=totals.device)
    s[: 2 * C] = totals
    return s


@pytest.mark.parametrize("M,N,K", [(4096, 256, 256), (100000, 128, 192), (3000, 1024, 1024), (50000, 64, 64)])
def test_gemm_nt_h_random_against_fp64(M, N, K):
    call, lib, ptr, _ = _api()
    gen = torch.Generator().manual_seed(M + N + K)
    a, w = torch.randn(M, K, generator=gen), torch.randn(N, K, generator=gen) / K ** 0.5
    a16, lda = _to16(a)
    w16, ldw = _to16(w)
    y = torch.empty(M, N, device=DEV)
    call("gemm_nt_h", ptr(a16), lda, ptr(w16), ldw, None, ptr(y), N, M, N, K, None, 0, 0)
    ar, wr = a.bfloat16().double(), w.bfloat16().double()
    want, bound = ar @ wr.t(), ar.abs() @ wr.abs().t()
    err = float(((y.cpu().double() - want).abs() / bound).max())
    assert err < 3e-6, err             # fp32 accumulation of exact bf16 products


@pytest.mark.parametrize("M,N,K", [(64, 128, 128), (1000, 128, 128), (777, 256, 192), (5000, 64, 320), (200000, 256, 256),
                                   (9000, 259, 131)])
def test_gemm_tn_h_exact_integers(M, N, K):
    """dW += dY^T X with transposed LDS reads: exact integer data (asymmetric in m, n and k), accumulation into a non-zero
    dW, rows beyond M, columns beyond N / K."""
    call, lib, ptr, workspace = _api()
    dy, x = _int_matrix(M, N, 3, -2, 3), _int_matrix(M, K, 4, -2, 3)
    # keep the exact sums inside fp32's integer range
    scale = max(1, int((M * 36) // 2 ** 23) + 1)
    if scale > 1:
        keep = (torch.arange(M) % scale == 0).float()[:, None]
        dy = dy * keep
    dy16, lddy = _to16(dy)
    x16, ldx = _to16(x)
    want = dy.double().t() @ x.double()
    assert float(want.abs().max()) < 2 ** 23
    base = (torch.arange(N * K).view(N, K) % 11).float()
    ldw = K + 3
    dw = torch.full((N, ldw), -2.0, device=DEV)
    dw[:, :K] = base.to(DEV)
    nb = lib().ccn_gemm_tn_h_workspace_bytes(M, N, K)
    ws = workspace(nb, DEV)
    call("gemm_tn_h", ptr(dy16), lddy, ptr(x16), ldx, ptr(dw), ldw, M, N, K, ptr(ws), nb)
    assert torch.equal(dw[:, :K].cpu().double(), want + base.double())
    assert bool((dw[:, K:] == -2.0).all())


@pytest.mark.parametrize("M,N,K", [(100000, 256, 256), (33333, 128, 64), (8000, 1024, 512)])
def test_gemm_tn_h_random_against_fp64(M, N, K):
    call, lib, ptr, workspace = _api()
    gen = torch.Generator().manual_seed(M + N + K)
    dy, x = torch.randn(M, N, generator=gen), torch.randn(M, K, generator=gen)
    dy16, lddy = _to16(dy)
    x16, ldx = _to16(x)
    dw = torch.zeros(N, K, device=DEV)
    nb = lib().ccn_gemm_tn_h_workspace_bytes(M, N, K)
    ws = workspace(nb, DEV)
    call("gemm_tn_h", ptr(dy16), lddy, ptr(x16), ldx, ptr(dw), K, M, N, K, ptr(ws), nb)
    a, b = dy.bfloat16().double(), x.bfloat16().double()
    want, bound = a.t() @ b, a.abs().t() @ b.abs()
    err = float(((dw.cpu().double() - want).abs() / bound).max())
    assert err < 3e-6, err
    # deterministic: the slabs are summed in chunk order
    dw2 = torch.zeros(N, K, device=DEV)
    call("gemm_tn_h", ptr(dy16), lddy, ptr(x16), ldx, ptr(dw2), K, M, N, K, ptr(ws), nb)
    assert torch.equal(dw, dw2)


@pytest.mark.parametrize("N,K", [(128, 256), (67, 131), (1024, 40)])
def test_transpose_cast(N, K):
    call, lib, ptr, _ = _api()
    gen = torch.Generator().manual_seed(N + K)
    w = torch.randn(N, K + 2, generator=gen).to(DEV)
    wt, ldt = _rows16(K, N)
    wt.fill_(3.0)
    call("transpose_cast_h", ptr(w), K + 2, N, K, ptr(wt), ldt, 0)
    assert torch.equal(wt[:, :N], w[:, :K].t().to(torch.bfloat16))
    assert bool((wt[:, N:] == 0).all())


@pytest.mark.parametrize("rows,C", [(1000, 64), (4099, 259), (300, 8), (70000, 128)])
@pytest.mark.parametrize("act", [1, 2])
def test_batchnorm_passes_with_16bit_rows(rows, C, act):
    """ccn_bn_act_fwd_h = ccn_bn_act_fwd followed by ONE rounding; ccn_bn_act_bwd_reduce_h / _apply_h = the fp32 passes on
    the same (bf16-valued) dZ, dY rounded once at the store."""
    call, lib, ptr, _ = _api()
    gen = torch.Generator().manual_seed(rows + C + act)
    y = torch.randn(rows, C, generator=gen).to(DEV)
    par = torch.stack([torch.rand(C, generator=gen) + 0.5, torch.randn(C, generator=gen) * 0.3,
                       torch.randn(C, generator=gen) * 0.1, torch.rand(C, generator=gen) + 0.5]).to(DEV)   # scale shift mean rstd
    z32 = torch.empty(rows, C, device=DEV)
    call("bn_act_fwd", ptr(y), C, rows, C, ptr(par[0]), ptr(par[1]), act, 0.01, ptr(z32), C)
    for dtype, f16 in ((torch.bfloat16, 0), (torch.float16, 1)):
        z16, ldz = _rows16(rows, C, dtype)
        z16.fill_(5.0)
        call("bn_act_fwd_h", ptr(y), C, rows, C, ptr(par[0]), ptr(par[1]), act, 0.01, ptr(z16), ldz, f16)
        assert torch.equal(z16[:, :C], z32.to(dtype))
        assert bool((z16[:, C:] == 0).all())
    # backward: dZ held in bf16
    g16, ldg = _to16(torch.randn(rows, C, generator=gen))
    g32 = g16[:, :C].float().contiguous()
    nparts = lib().ccn_stats_rows(rows)
    s_ref = torch.zeros((nparts + 1) * 2 * C, dtype=torch.float64, device=DEV)
    s_h = torch.zeros_like(s_ref)
    call("bn_act_bwd_reduce", ptr(g32), C, ptr(y), C, rows, C, ptr(par[0]), ptr(par[1]), ptr(par[2]), ptr(par[3]), act, 0.01,
         ptr(s_ref))
    call("bn_act_bwd_reduce_h", ptr(g16), ldg, ptr(y), C, rows, C, ptr(par[0]), ptr(par[1]), ptr(par[2]), ptr(par[3]), act,
         0.01, ptr(s_h))
    tot_ref, tot_h = s_ref[: 2 * C].cpu(), s_h[: 2 * C].cpu()
    assert torch.allclose(tot_h, tot_ref, rtol=1e-5, atol=1e-4 * rows ** 0.5), float((tot_h - tot_ref).abs().max())
    dy32 = torch.empty(rows, C, device=DEV)
    dgb = torch.zeros(2, C, device=DEV)
    call("bn_act_bwd_apply_ex", ptr(g32), C, ptr(y), C, rows, C, ptr(par[0]), ptr(par[1]), ptr(par[2]), ptr(par[3]), act, 0.01,
         ptr(s_ref), float(rows), 1, 0, ptr(dy32), C, ptr(dgb[0]), ptr(dgb[1]))
    for dz16, gsrc, ldsrc in ((1, g16, ldg), (0, g32, C)):
        dy16, lddy = _rows16(rows, C)
        dgb2 = torch.full((2, C), 9.0, device=DEV)
        call("bn_act_bwd_apply_h", ptr(gsrc), dz16, ldsrc, ptr(y), C, rows, C, ptr(par[0]), ptr(par[1]), ptr(par[2]),
             ptr(par[3]), act, 0.01, ptr(s_ref), float(rows), 1, 0, ptr(dy16), lddy, ptr(dgb2[0]), ptr(dgb2[1]), 0)
        # the same expression, but the two kernels may contract multiply-adds differently: equal up to one bf16 rounding
        # boundary crossed by a last-bit fp32 difference
        diff = (dy16[:, :C].float() - dy32.to(torch.bfloat16).float()).abs()
        tol = dy32.abs() * 2 ** -7 + 2e-6            # (+ fp32 cancellation noise of g - m1 - xhat m2 where it is ~0)
        assert bool((diff <= tol).all())
        assert float((diff > 0).float().mean()) < 1e-3
        assert torch.equal(dgb2, dgb)
        assert bool((dy16[:, C:] == 0).all())


# ---------------------------------------------------------------- the MLP boundaries of the 16-bit storage modes (round 3)
@pytest.mark.parametrize("rows,C", [(1000, 64), (4099, 72), (33, 8), (70001, 128)])
def test_add_cast_rows(rows, C):
    """ccn_add_cast_rows_h: bf16(fp32 gradient + bf16 gradient), one rounding of the fp32 sum."""
    call, lib, ptr, _ = _api()
    gen = torch.Generator().manual_seed(rows)
    a = torch.randn(rows, C, generator=gen).to(DEV)
    b16, ldb = _to16(torch.randn(rows, C, generator=gen))
    out, ldo = _rows16(rows, C)
    out.fill_(3.0)
    call("add_cast_rows_h", ptr(a), C, ptr(b16), ldb, rows, C, ptr(out), ldo)
    assert torch.equal(out[:, :C], (a + b16[:, :C].float()).to(torch.bfloat16))
    assert bool((out[:, C:] == 0).all())


@pytest.mark.parametrize("rows,C,taps", [(500, 16, 5), (3000, 64, 5), (257, 8, 7)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_im2col_16bit_rows(rows, C, taps, dtype):
    """ccn_im2col_fwd_h = ccn_im2col_fwd rounded once; ccn_im2col_bwd_h = ccn_im2col_bwd on the widened bf16 gradient."""
    call, lib, ptr, _ = _api()
    gen = torch.Generator().manual_seed(rows + taps)
    x = torch.randn(rows, C, generator=gen).to(DEV)
    seg = torch.sort(torch.randint(0, 9, (rows,), generator=gen)).values.to(torch.int32).to(DEV)
    W = taps * C
    col32 = torch.empty(rows, W, device=DEV)
    call("im2col_fwd", ptr(x), C, ptr(seg), rows, C, taps, ptr(col32), W)
    col16, ld = _rows16(rows, W, dtype)
    call("im2col_fwd_h", ptr(x), C, ptr(seg), rows, C, taps, ptr(col16), ld, 1 if dtype == torch.float16 else 0)
    assert torch.equal(col16[:, :W], col32.to(dtype))
    g16, ldg = _to16(torch.randn(rows, W, generator=gen))
    g32 = g16[:, :W].float().contiguous()
    dx_a, dx_b = torch.empty(rows, C, device=DEV), torch.empty(rows, C, device=DEV)
    call("im2col_bwd", ptr(g32), W, ptr(seg), rows, C, taps, ptr(dx_a), C)
    call("im2col_bwd_h", ptr(g16), ldg, ptr(seg), rows, C, taps, ptr(dx_b), C)
    assert torch.equal(dx_a, dx_b)


def _random_groups(n_groups, max_len, gen):
    lens = torch.randint(0, max_len + 1, (n_groups,), generator=gen)
    offsets = torch.zeros(n_groups + 1, dtype=torch.int64)
    offsets[1:] = torch.cumsum(lens, 0)
    return offsets.to(torch.int32), int(offsets[-1])


@pytest.mark.parametrize("n_dst,C", [(700, 64), (90, 136), (3000, 8)])
def test_softmax_aggregation_backward_writes_bf16_scores_gradient(n_dst, C):
    """ccn_seg_softmax_agg_bwd_h: the messages' gradient bit-identical to the fp32 entry, the scores' gradient = its fp32
    value rounded once (empty groups included)."""
    call, lib, ptr, _ = _api()
    gen = torch.Generator().manual_seed(n_dst)
    offsets, e = _random_groups(n_dst, 12, gen)
    offsets = offsets.to(DEV)
    msg, att = torch.randn(e, C, generator=gen).to(DEV), torch.randn(e, C, generator=gen).to(DEV)
    g = torch.randn(n_dst, C, generator=gen).to(DEV)
    dm_a, da_a = torch.empty(e, C, device=DEV), torch.empty(e, C, device=DEV)
    call("seg_softmax_agg_bwd", ptr(msg), C, ptr(att), C, ptr(offsets), n_dst, C, ptr(g), C, ptr(dm_a), C, ptr(da_a), C)
    dm_b = torch.empty(e, C, device=DEV)
    da_b, ld = _rows16(e, C)
    call("seg_softmax_agg_bwd_h", ptr(msg), C, ptr(att), C, ptr(offsets), n_dst, C, ptr(g), C, ptr(dm_b), C, ptr(da_b), ld)
    assert torch.equal(dm_a, dm_b)
    assert torch.equal(da_b[:, :C], da_a.to(torch.bfloat16))


@pytest.mark.parametrize("n,C", [(500, 64), (77, 72), (2000, 8)])
def test_compact_row_max_backward_writes_bf16_rows(n, C):
    """ccn_cg_max_bwd_h against ccn_cg_max_bwd on a random compact-row structure (real rows grouped by point, optional
    representative row per point, one padding row): every row of the table written, same values rounded once."""
    call, lib, ptr, _ = _api()
    gen = torch.Generator().manual_seed(n + C)
    lens = torch.randint(1, 9, (n,), generator=gen)                      # the self row always exists
    grp = torch.zeros(n + 1, dtype=torch.int64)
    grp[1:] = torch.cumsum(lens, 0)
    e = int(grp[-1])
    has_rep = torch.rand(n, generator=gen) < 0.4
    rep = torch.full((n,), -1, dtype=torch.int64)
    rep[has_rep] = e + torch.arange(int(has_rep.sum()))
    ne = int(has_rep.sum())
    rows = e + ne + 1
    arg = (torch.rand(n, C, generator=gen) * lens[:, None].float()).long().clamp(max=8).to(torch.int32)
    arg = torch.minimum(arg, (lens[:, None] - 1).to(torch.int32))
    g = torch.randn(n, C, generator=gen).to(DEV)
    grp_d, rep_d, arg_d = grp.to(torch.int32).to(DEV), rep.to(torch.int32).to(DEV), arg.contiguous().to(DEV)
    df32 = torch.full((rows, C), 5.0, device=DEV)
    call("cg_max_bwd", ptr(g), C, ptr(arg_d), ptr(grp_d), ptr(rep_d), n, rows, C, ptr(df32), C)
    df16, ld = _rows16(rows, C)
    df16.fill_(5.0)
    call("cg_max_bwd_h", ptr(g), C, ptr(arg_d), ptr(grp_d), ptr(rep_d), n, rows, C, ptr(df16), ld)
    assert torch.equal(df16[:, :C], df32.to(torch.bfloat16))
    assert float(df32.abs().sum()) > 0 and bool((df32[e:] == 0).all())
