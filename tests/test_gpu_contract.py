"""The SURVEY section 8(b) entry points of the C-ABI (ccn_contract.hip), called the way a non-Python host would: raw
pointers, caller-owned workspaces.  Checked against plain torch fp32 references (F.conv1d, F.linear + F.batch_norm,
index gathers) and, where the Python mirror runs the same kernels in the same order, for bit equality with it."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

from tests.util import maxdiff

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
LEAKY, RELU = 2, 1


def _api():
    from curvecloudnet_amd import ops
    from curvecloudnet_amd._lib import call, lib, ptr, workspace
    return ops, call, lib, ptr, workspace


def _close(a, b, tol, what):
    scale = max(1.0, float(b.abs().max()))
    err = maxdiff(a, b)
    assert err <= tol * scale, "%s: max |diff| %.3g (scale %.3g)" % (what, err, scale)


def _seq(x, h, ld):
    """(L, C) -> ((L + 2h) x ld buffer with zero halo rows and zero padding columns, view of the L real rows)."""
    buf = torch.zeros(x.size(0) + 2 * h, ld, device=DEV)
    buf[h:h + x.size(0), :x.size(1)] = x.to(DEV)
    return buf


@pytest.mark.parametrize("L,cin,cout,taps", [(5000, 8, 32, 5), (20000, 32, 32, 5), (3000, 134, 64, 7), (9000, 262, 128, 5),
                                             (700, 3, 5, 3)])
def test_curve_conv_entries_vs_conv1d(L, cin, cout, taps):
    ops, call, lib, ptr, workspace = _api()
    gen = torch.Generator().manual_seed(L + cin)
    h = taps // 2
    x = torch.randn(L, cin, generator=gen)
    w = torch.randn(cout, cin, taps, generator=gen) / (cin * taps) ** 0.5
    b = torch.randn(cout, generator=gen)
    cot = torch.randn(L, cout, generator=gen)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = F.conv1d(xr.t()[None], wr, b, padding=h)[0].t()
    gx, gw = torch.autograd.grad((yr * cot).sum(), [xr, wr])

    ld = (cin + 3) // 4 * 4
    ldo = (cout + 3) // 4 * 4
    seq = _seq(x, h, ld)
    wg = torch.zeros(cout, taps, ld, device=DEV)
    wg[:, :, :cin] = w.permute(0, 2, 1).to(DEV)                      # [tap][channel]
    y = torch.empty(L, ldo, device=DEV)
    stats = torch.empty((lib().ccn_stats_rows(L) + 1) * 2 * cout, dtype=torch.float64, device=DEV)
    bd = b.to(DEV)
    call("curve_conv_fwd", ptr(seq), ld, L, taps, ptr(wg), taps * ld, ptr(bd), cout, ptr(y), ldo, ptr(stats))
    _close(y[:, :cout].cpu(), yr.detach(), 3e-5, "conv fwd")
    nparts = lib().ccn_stats_rows(L)
    tot = stats.view(nparts + 1, 2, cout)[:nparts].sum(0).cpu()
    _close(tot[0].float(), yr.detach().double().sum(0).float(), 1e-4, "column sums")
    _close(tot[1].float(), (yr.detach().double() ** 2).sum(0).float(), 1e-4, "column sums of squares")

    dyseq = _seq(cot, h, ldo)
    dx = torch.empty(L, ld, device=DEV)
    nb = lib().ccn_curve_conv_bwd_data_workspace_bytes(cin, taps, ldo)
    ws = workspace(nb, DEV)
    call("curve_conv_bwd_data", ptr(dyseq), ldo, L, taps, ptr(wg), ld, cout, cin, ptr(dx), ld, ptr(ws), ws.numel())
    _close(dx[:, :cin].cpu(), gx, 5e-5, "conv data gradient")

    dw = torch.zeros(cout, taps * ld, device=DEV)
    nb = lib().ccn_curve_conv_bwd_weight_workspace_bytes(L, cout, taps, ld)
    ws = workspace(nb, DEV)
    dy0 = ctypes.c_void_p(dyseq.data_ptr() + h * ldo * 4)
    for _ in range(2):                                               # accumulates: twice = 2 x
        call("curve_conv_bwd_weight", dy0, ldo, ptr(seq), ld, L, taps, cout, ptr(dw), taps * ld, ptr(ws), ws.numel())
    got = dw.view(cout, taps, ld)[:, :, :cin].permute(0, 2, 1).cpu()
    _close(got, 2 * gw, 1e-4, "conv weight gradient")
    assert float(dw.view(cout, taps, ld)[:, :, cin:].abs().max()) == 0.0 if ld > cin else True


@pytest.mark.parametrize("rows,c,act,training", [(4000, 64, LEAKY, 1), (1237, 259, RELU, 1), (900, 32, LEAKY, 0)])
def test_bn_act_bwd_entry_vs_torch(rows, c, act, training):
    ops, call, lib, ptr, workspace = _api()
    gen = torch.Generator().manual_seed(rows)
    y = torch.randn(rows, c, generator=gen) * 2 + 0.5
    gamma = torch.rand(c, generator=gen) + 0.5
    beta = torch.randn(c, generator=gen)
    rm, rv = torch.randn(c, generator=gen) * 0.1, torch.rand(c, generator=gen) + 0.5
    cot = torch.randn(rows, c, generator=gen)
    yr, gr, br = (v.clone().requires_grad_(True) for v in (y, gamma, beta))
    zn = F.batch_norm(yr, rm.clone(), rv.clone(), gr, br, bool(training), 0.1, 1e-5)
    zr = F.leaky_relu(zn, 0.01) if act == LEAKY else F.relu(zn)
    gy, gg, gb = torch.autograd.grad((zr * cot).sum(), [yr, gr, br])

    ld = (c + 3) // 4 * 4
    yd = torch.zeros(rows, ld, device=DEV)
    yd[:, :c] = y.to(DEV)
    par = torch.empty(4, c, device=DEV)
    g_, b_, rm_, rv_ = (v.to(DEV) for v in (gamma, beta, rm, rv))
    if training:
        # (scale, shift, mean, rstd) as ccn_bn_finalize leaves them, from an fp64 pass over the batch
        mean, var = y.double().mean(0), y.double().var(0, unbiased=False)
        rstd = (var + 1e-5).rsqrt()
        par[0] = (gamma.double() * rstd).float().to(DEV)
        par[1] = (beta.double() - mean * gamma.double() * rstd).float().to(DEV)
        par[2], par[3] = mean.float().to(DEV), rstd.float().to(DEV)
    else:
        call("bn_eval_params", ptr(g_), ptr(b_), ptr(rm_), ptr(rv_), 1e-5, c, ptr(par[0]), ptr(par[1]), ptr(par[2]), ptr(par[3]))
    dz = torch.zeros(rows, ld, device=DEV)
    dz[:, :c] = cot.to(DEV)
    dy = torch.empty(rows, ld, device=DEV)
    dgb = torch.empty(2, c, device=DEV)
    ws = workspace(lib().ccn_bn_act_bwd_workspace_bytes(rows, c), DEV)
    call("bn_act_bwd", ptr(dz), ld, ptr(yd), ld, rows, c, ptr(par), act, 0.01, training, ptr(dy), ld, ptr(dgb[0]), ptr(dgb[1]),
         ptr(ws), ws.numel())
    _close(dy[:, :c].cpu(), gy, 5e-5, "dY")
    _close(dgb[0].cpu(), gg, 1e-4, "dgamma")
    _close(dgb[1].cpu(), gb, 1e-4, "dbeta")


def _layer_ref(x, w, b, gamma, beta, rm, rv, training, act, cot, route=None):
    """fp64 on the CPU; running statistics updated in place as F.batch_norm does.  ``route`` (rows x N bool): the branch of
    the activation the GPU took per element -- a normalised value within rounding of the kink may land on either side, and one
    flipped ReLU moves a row of dX by ~0.1 and an entry of dW by ~|x|; the reference then differentiates the same branch."""
    d = lambda v: None if v is None else v.double().clone().requires_grad_(True)
    xr, wr, br, gr, ber = d(x), d(w), d(b), d(gamma), d(beta)
    leaves = [v for v in (xr, wr, br) if v is not None]
    y = F.linear(xr, wr, br)
    z = y
    if gamma is not None:
        leaves += [gr, ber]
        # written out (this torch build's CPU batch_norm backward is off by ~5 % from 40 000 rows on, in fp32 AND fp64;
        # its GPU kernel and this formula agree with each other)
        if training:
            mean, var = y.mean(0), y.var(0, unbiased=False)
            rm.mul_(0.9).add_(0.1 * mean.detach().float())
            rv.mul_(0.9).add_(0.1 * (var.detach() * y.size(0) / (y.size(0) - 1)).float())
        else:
            mean, var = rm.double(), rv.double()
        zn = (y - mean) / (var + 1e-5).sqrt() * gr + ber
        if route is None:
            z = F.leaky_relu(zn, 0.01) if act == LEAKY else F.relu(zn)
        else:
            z = zn * torch.where(route, 1.0, 0.01 if act == LEAKY else 0.0).double()
    grads = torch.autograd.grad((z * cot.double()).sum(), leaves)
    return y.detach().float(), z.detach().float(), [g.float() for g in grads]


@pytest.mark.parametrize("M,K,N,bias,bn,training,act", [
    (5000, 67, 64, False, True, True, LEAKY), (3001, 256, 128, True, True, True, RELU), (2000, 259, 128, False, True, False, RELU),
    (4096, 128, 20, True, False, False, 0), (130000, 256, 256, False, True, True, RELU)])
def test_linear_bn_act_entries(M, K, N, bias, bn, training, act):
    """fwd + bwd through ccn_linear_bn_act_* against torch, and bit-equal to the autograd function of the Python mirror."""
    ops, call, lib, ptr, workspace = _api()
    gen = torch.Generator().manual_seed(M + K + N)
    x = torch.randn(M, K, generator=gen)
    w = torch.randn(N, K, generator=gen) / K ** 0.5
    b = torch.randn(N, generator=gen) if bias else None
    gamma = torch.rand(N, generator=gen) + 0.5 if bn else None
    beta = torch.randn(N, generator=gen) if bn else None
    rm, rv = torch.randn(N, generator=gen) * 0.1, torch.rand(N, generator=gen) + 0.5
    cot = torch.randn(M, N, generator=gen)
    rm_ref, rv_ref = rm.clone(), rv.clone()

    ldx, ldn = (K + 3) // 4 * 4, (N + 3) // 4 * 4
    xd = torch.zeros(M, ldx, device=DEV)
    xd[:, :K] = x.to(DEV)
    wd = torch.zeros(N, ldx, device=DEV)
    wd[:, :K] = w.to(DEV)
    bd = b.to(DEV) if bias else None
    gd, bed = (gamma.to(DEV), beta.to(DEV)) if bn else (None, None)
    rmd, rvd = rm.to(DEV), rv.to(DEV)
    y = torch.empty(M, ldn, device=DEV)
    z = torch.empty(M, ldn, device=DEV) if bn else None
    par = torch.empty(4, N, device=DEV) if bn else None
    ws = workspace(lib().ccn_linear_bn_act_workspace_bytes(M, N, K), DEV)
    call("linear_bn_act_fwd", ptr(xd), ldx, ptr(wd), ldx, ptr(bd), ptr(gd), ptr(bed), ptr(rmd), ptr(rvd), M, N, K, 1e-5, 0.1,
         int(training), act, 0.01, 0, ptr(y), ldn, ptr(z), ldn, ptr(par), ptr(ws), ws.numel())
    route = (z[:, :N] > 0).cpu() if bn else None
    y_ref, z_ref, g_ref = _layer_ref(x, w, b, gamma, beta, rm_ref, rv_ref, training, act, cot, route)
    _close(y[:, :N].cpu(), y_ref, 3e-5, "product")
    if bn:
        _close(z[:, :N].cpu(), z_ref, 1e-4, "activation")
        if training:
            _close(rmd.cpu(), rm_ref, 1e-5, "running mean")
            _close(rvd.cpu(), rv_ref, 1e-5, "running var")
    dz = torch.zeros(M, ldn, device=DEV)
    dz[:, :N] = cot.to(DEV)
    dy = torch.empty(M, ldn, device=DEV) if bn else None
    dx = torch.empty(M, ldx, device=DEV)
    dw = torch.zeros(N, ldx, device=DEV)
    db = torch.empty(N, device=DEV) if bias else None
    dgb = torch.empty(2, N, device=DEV) if bn else None
    call("linear_bn_act_bwd", ptr(dz), ldn, ptr(xd), ldx, ptr(wd), ldx, ptr(y), ldn, ptr(par), M, N, K, int(training), act, 0.01,
         0, ptr(dy), ldn, ptr(dx), ldx, ptr(dw), ldx, ptr(db), ptr(dgb[0]) if bn else None, ptr(dgb[1]) if bn else None,
         ptr(ws), ws.numel())
    gl = list(g_ref)
    tol = 2e-4 if M > 50000 else 1e-4
    _close(dx[:, :K].cpu(), gl.pop(0), tol, "dX")
    _close(dw[:, :K].cpu(), gl.pop(0), tol, "dW")
    if bias:
        _close(db.cpu(), gl.pop(0), tol, "dbias")
    if bn:
        _close(dgb[0].cpu(), gl.pop(0), tol, "dgamma")
        _close(dgb[1].cpu(), gl.pop(0), tol, "dbeta")

    # the Python mirror runs the same kernels in the same order: same bits
    bnm = None
    if bn:
        bnm = torch.nn.BatchNorm1d(N).to(DEV)
        with torch.no_grad():
            bnm.weight.copy_(gd), bnm.bias.copy_(bed), bnm.running_mean.copy_(rm.to(DEV)), bnm.running_var.copy_(rv.to(DEV))
    xa = xd[:, :K].detach().requires_grad_(True)
    wa = wd[:, :K].detach().requires_grad_(True)
    ba = bd.detach().requires_grad_(True) if bias else None
    za = ops.linear_bn_act(xa, wa, ba, bnm, bool(training), {LEAKY: "leaky_relu", RELU: "relu", 0: None}[act])
    assert torch.equal(za, (z if bn else y)[:, :N]), "forward differs from the Python mirror"
    leaves = [xa, wa] + ([ba] if bias else []) + ([bnm.weight, bnm.bias] if bn else [])
    ga = torch.autograd.grad((za * dz[:, :N]).sum(), leaves)
    assert torch.equal(ga[0], dx[:, :K]), "dX differs from the Python mirror"
    if N >= 64 and K >= 64:
        assert torch.equal(ga[1], dw[:, :K]), "dW differs from the Python mirror"
    else:            # narrow shapes take the split-K kernel, whose atomic order is not fixed
        _close(ga[1].cpu(), dw[:, :K].cpu(), 1e-5, "dW vs the Python mirror")


@pytest.mark.parametrize("dtype,name", [(1, "bf16"), (2, "fp16")])
def test_linear_bn_act_entries_16bit(dtype, name):
    """The 16-bit MLP modes through the contract entry: same bits as the Python mirror in that mode."""
    ops, call, lib, ptr, workspace = _api()
    M, K, N = 6000, 128, 64
    gen = torch.Generator().manual_seed(dtype)
    xd = torch.randn(M, K, generator=gen).to(DEV)
    wd = (torch.randn(N, K, generator=gen) / K ** 0.5).to(DEV)
    cot = torch.randn(M, N, generator=gen).to(DEV)
    bnm = torch.nn.BatchNorm1d(N).to(DEV)
    rmd, rvd = bnm.running_mean.clone(), bnm.running_var.clone()
    y, z, par = torch.empty(M, N, device=DEV), torch.empty(M, N, device=DEV), torch.empty(4, N, device=DEV)
    ws = workspace(lib().ccn_linear_bn_act_workspace_bytes(M, N, K), DEV)
    call("linear_bn_act_fwd", ptr(xd), K, ptr(wd), K, None, ptr(bnm.weight), ptr(bnm.bias), ptr(rmd), ptr(rvd), M, N, K, 1e-5,
         0.1, 1, RELU, 0.01, dtype, ptr(y), N, ptr(z), N, ptr(par), ptr(ws), ws.numel())
    dy, dx, dw, dgb = torch.empty(M, N, device=DEV), torch.empty(M, K, device=DEV), torch.zeros(N, K, device=DEV), torch.empty(2, N, device=DEV)
    call("linear_bn_act_bwd", ptr(cot), N, ptr(xd), K, ptr(wd), K, ptr(y), N, ptr(par), M, N, K, 1, RELU, 0.01, dtype, ptr(dy),
         N, ptr(dx), K, ptr(dw), K, None, ptr(dgb[0]), ptr(dgb[1]), ptr(ws), ws.numel())
    prev = ops.mlp_dtype()
    ops.set_mlp_dtype(name)
    try:
        # the contract entry composes the fp32-STORAGE kernels: the Python mirror in that form gives the same bits ...
        ops.STORE16 = False
        xa, wa = xd.clone().requires_grad_(True), wd.clone().requires_grad_(True)
        za = ops.linear_bn_act(xa, wa, None, bnm, True, "relu")
        ga = torch.autograd.grad((za * cot).sum(), [xa, wa])
        # ... and the 16-bit storage form (ops.STORE16, the default of the bf16 mode) the same values up to the fp32
        # summation order of its kernels (and, in the gradients, bf16 rounding boundaries crossed by that difference)
        ops.STORE16 = True
        xs, wsd = xd.clone().requires_grad_(True), wd.clone().requires_grad_(True)
        zs = ops.linear_bn_act(xs, wsd, None, bnm, True, "relu")
        gs = torch.autograd.grad((zs * cot).sum(), [xs, wsd])
    finally:
        ops.STORE16 = True
        ops.set_mlp_dtype(prev)
    assert torch.equal(za, z) and torch.equal(ga[0], dx)
    _close(ga[1].cpu(), dw.cpu(), 1e-5, "dW vs the Python mirror (split-K atomics: order not fixed)")
    _close(zs.cpu(), z.cpu(), 1e-4, "16-bit storage form: activation")
    _close(gs[0].cpu(), dx.cpu(), 2e-2, "16-bit storage form: dx")
    _close(gs[1].cpu(), dw.cpu(), 2e-2, "16-bit storage form: dW")
    y32 = xd @ wd.t()
    assert maxdiff(y.cpu(), y32.cpu()) < 5e-2 and maxdiff(y.cpu(), y32.cpu()) > 0          # 16-bit products, not fp32


def test_gather_and_reduce_entries_alias_the_kernels():
    ops, call, lib, ptr, workspace = _api()
    gen = torch.Generator().manual_seed(3)
    b, nmax, k, c = 2, 300, 8, 48
    lengths = torch.tensor([300, 211])
    cloud_ptr = torch.tensor([0, 300, 511]).to(DEV)
    x = torch.randn(511, c, generator=gen).to(DEV)
    idx = torch.full((b, nmax, k), -1, dtype=torch.int64)
    for i in range(b):
        n = int(lengths[i])
        idx[i, :n] = torch.randint(0, n, (n, k), generator=gen)
        idx[i, :n, -1] = -1
    idx = idx.to(DEV)
    f1 = torch.empty(b * nmax * k, c, device=DEV)
    call("gather_edge_fwd", ptr(x), c, ptr(idx), ptr(cloud_ptr), b, nmax, k, c, ptr(f1), c)
    # frnn_gather semantics: per cloud x[idx], zero where idx < 0 and in the rows past the cloud's length
    xp = torch.zeros(b, nmax, c, device=DEV)
    xp[0, :300], xp[1, :211] = x[:300], x[300:]
    ref = torch.gather(xp[:, :, None].expand(-1, -1, k, -1), 1, idx.clamp_min(0)[..., None].expand(-1, -1, -1, c))
    ref = ref * (idx >= 0)[..., None]
    assert torch.equal(f1.view(b, nmax, k, c), ref)
    cot = torch.randn(b * nmax * k, c, generator=gen).to(DEV)
    d1 = torch.zeros(511, c, device=DEV)
    call("gather_edge_bwd", ptr(cot), c, ptr(idx), ptr(cloud_ptr), b, nmax, k, c, ptr(d1), c)
    xg = x.clone().requires_grad_(True)
    xq = torch.zeros(b, nmax, c, device=DEV)
    xq[0, :300], xq[1, :211] = xg[:300], xg[300:]
    rg = torch.gather(xq[:, :, None].expand(-1, -1, k, -1), 1, idx.clamp_min(0)[..., None].expand(-1, -1, -1, c))
    (gxr,) = torch.autograd.grad(((rg * (idx >= 0)[..., None]).reshape(-1, c) * cot).sum(), xg)
    _close(d1.cpu(), gxr.cpu(), 1e-5, "gather bwd")

    m, e = 400, 3000
    cnt = torch.randint(1, 14, (m,), generator=gen)
    offsets = torch.cat([torch.zeros(1, dtype=torch.int64), cnt.cumsum(0)]).to(torch.int32).to(DEV)
    e = int(offsets[-1])
    msg, att = torch.randn(e, c, generator=gen).to(DEV), torch.randn(e, c, generator=gen).to(DEV)
    o1, o2 = torch.empty(m, c, device=DEV), torch.empty(m, c, device=DEV)
    a1, a2 = torch.empty(m, c, dtype=torch.int32, device=DEV), torch.empty(m, c, dtype=torch.int32, device=DEV)
    call("edge_reduce_max_fwd", ptr(msg), c, ptr(offsets), m, c, ptr(o1), c, ptr(a1))
    call("seg_max_fwd", ptr(msg), c, ptr(offsets), m, c, ptr(o2), c, ptr(a2))
    assert torch.equal(o1, o2) and torch.equal(a1, a2)
    seg = torch.repeat_interleave(torch.arange(m), cnt).to(DEV)
    ref = torch.full((m, c), -float("inf"), device=DEV).scatter_reduce(0, seg[:, None].expand(-1, c), msg, "amax")
    assert torch.equal(o1, ref)
    call("edge_reduce_attend_fwd", ptr(msg), c, ptr(att), c, ptr(offsets), m, c, ptr(o1), c)
    call("seg_softmax_agg_fwd", ptr(msg), c, ptr(att), c, ptr(offsets), m, c, ptr(o2), c)
    assert torch.equal(o1, o2)
    mx = torch.full((m, c), -float("inf"), device=DEV).scatter_reduce(0, seg[:, None].expand(-1, c), att, "amax")
    ex = (att - mx[seg]).exp()
    den = torch.zeros(m, c, device=DEV).index_add_(0, seg, ex) + 1e-16
    ref = torch.zeros(m, c, device=DEV).index_add_(0, seg, ex / den[seg] * msg)
    _close(o1.cpu(), ref.cpu(), 2e-5, "attend aggregation")


@pytest.mark.parametrize("M,N,K", [(40000, 192, 128), (20000, 256, 256), (33000, 128, 64), (17000, 256, 192), (9000, 1024, 1024)])
@pytest.mark.parametrize("act", [RELU, LEAKY])
def test_transforming_gemms_match_written_activation(M, N, K, act):
    """ccn_gemm_nt_xf / ccn_gemm_tn_ws_xf (the previous layer's BatchNorm + activation applied to the operand fragments
    inside the kernel) give the bits of ccn_bn_act_fwd followed by ccn_gemm_nt / ccn_gemm_tn_ws, statistics included."""
    ops, call, lib, ptr, workspace = _api()
    gen = torch.Generator().manual_seed(M + N + K + act)
    y0 = torch.randn(M, K, generator=gen).to(DEV)
    w = (torch.randn(N, K, generator=gen) / K ** 0.5).to(DEV)
    b = torch.randn(N, generator=gen).to(DEV)
    par = torch.stack([torch.rand(K, generator=gen) + 0.5, torch.randn(K, generator=gen)]).to(DEV)
    dy = torch.randn(M, N, generator=gen).to(DEV)
    z = torch.empty(M, K, device=DEV)
    call("bn_act_fwd", ptr(y0), K, M, K, ptr(par[0]), ptr(par[1]), act, 0.01, ptr(z), K)
    zr = y0 * par[0] + par[1]
    zr = torch.where(zr > 0, zr, zr * (0.01 if act == LEAKY else 0.0))
    _close(z.cpu(), zr.cpu(), 1e-6, "bn_act_fwd")
    nparts = lib().ccn_stats_rows(M)
    s1 = torch.empty((nparts + 1) * 2 * N, dtype=torch.float64, device=DEV)
    s2 = torch.empty_like(s1)
    o1, o2 = torch.empty(M, N, device=DEV), torch.empty(M, N, device=DEV)
    call("gemm_nt", ptr(z), K, ptr(w), K, ptr(b), ptr(o1), N, M, N, K, ptr(s1))
    if lib().ccn_gemm_nt_xf_ok(K, K, M, N, K):
        call("gemm_nt_xf", ptr(y0), K, ptr(par[0]), ptr(par[1]), act, 0.01, ptr(w), K, ptr(b), ptr(o2), N, M, N, K, ptr(s2))
        assert torch.equal(o1, o2), "product"
        assert torch.equal(s1[:nparts * 2 * N], s2[:nparts * 2 * N]), "statistics"
    else:
        assert K > 1024 or N <= 64 or (M // 128) * (N // 128) < 128, "shape unexpectedly outside the fused kernel"
    nb = lib().ccn_gemm_tn_workspace_bytes(M, N, K)
    ws = workspace(nb, DEV)
    d1, d2 = torch.zeros(N, K, device=DEV), torch.zeros(N, K, device=DEV)
    call("gemm_tn_ws", ptr(dy), N, ptr(z), K, ptr(d1), K, M, N, K, ptr(ws), nb)
    assert lib().ccn_gemm_tn_xf_ok(ptr(dy), N, ptr(y0), K, M, N, K)
    call("gemm_tn_ws_xf", ptr(dy), N, ptr(y0), K, ptr(par[0]), ptr(par[1]), act, 0.01, ptr(d2), K, M, N, K, ptr(ws), nb)
    assert torch.equal(d1, d2), "weight gradient"


@pytest.mark.parametrize("M,N,K", [(2051, 130, 96), (4099, 131, 259), (1033, 192, 67)])
def test_weight_gradient_remainder_block_on_a_tight_allocation(M, N, K):
    """ccn_gemm_tn_ws splits an output width like 130 / 131 / 259 into a 128-multiple block and a remainder block whose
    operand pointers are shifted by n0 / k0.  The remainder block's LDS-DMA columns must be clamped against what is left
    of the row (ld - n0), not against ld (up to 240 B past the end of dY on the last row).  The operands sit at
    the very END of their allocations, with a sentinel buffer behind them that must stay untouched, and the result is
    held against an fp64 product."""
    ops, call, lib, ptr, workspace = _api()
    gen = torch.Generator().manual_seed(M + N + K)
    ldn, ldk = (N + 3) // 4 * 4, (K + 3) // 4 * 4
    pool = torch.full((M * ldn + M * ldk + 4096,), float("nan"), device=DEV)       # [guard | dY | X], both flush right
    dy = pool[4096:4096 + M * ldn].view(M, ldn)
    x = pool[4096 + M * ldn:].view(M, ldk)
    dy.copy_(torch.randn(M, ldn, generator=gen))
    x.copy_(torch.randn(M, ldk, generator=gen))
    assert x.data_ptr() + x.numel() * 4 == pool.data_ptr() + pool.numel() * 4       # X ends with the allocation
    dw = torch.zeros(N, ldk, device=DEV)
    nb = lib().ccn_gemm_tn_workspace_bytes(M, N, K)
    ws = workspace(max(nb, 16), DEV)
    call("gemm_tn_ws", ptr(dy), ldn, ptr(x), ldk, ptr(dw), ldk, M, N, K, ptr(ws), nb)
    torch.cuda.synchronize()
    ref = dy[:, :N].double().t() @ x[:, :K].double()
    bound = (dy[:, :N].double().abs().t() @ x[:, :K].double().abs())
    err = float(((dw[:, :K].double() - ref).abs() / bound).max())
    assert err < 5e-6, err
    assert bool((dw[:, K:] == 0).all())                                              # padding columns of dW untouched


@pytest.mark.parametrize("M,N,K,act", [(40000, 256, 256, 1), (33000, 192, 128, 2), (20000, 259, 512, 2), (16500, 128, 1027, 0)])
def test_gemm_nt_red_equals_product_plus_reduce(M, N, K, act):
    """ccn_gemm_nt_red (round 4): the data-gradient product with the previous layer's BatchNorm-backward column sums taken in
    its epilogue == ccn_gemm_nt followed by ccn_bn_act_bwd_reduce over (dZ, y): same dZ bits, sums to fp32-summation accuracy.
    Ragged last tiles (M, N not multiples of 128), a K remainder, every activation code."""
    from curvecloudnet_amd._lib import lib, ptr
    gen = torch.Generator().manual_seed(M + N)
    lda = (K + 3) // 4 * 4
    a = torch.randn(M, lda, generator=gen).to(DEV)
    w = (torch.randn(N, lda, generator=gen) / K ** 0.5).to(DEV)
    ldy = (N + 3) // 4 * 4 + 4
    y_prev = torch.randn(M, ldy, generator=gen).to(DEV)
    par = torch.stack([torch.rand(N, generator=gen) + 0.5, torch.randn(N, generator=gen) * 0.3,
                       torch.randn(N, generator=gen) * 0.2, torch.rand(N, generator=gen) + 0.5]).to(DEV).contiguous()
    assert lib().ccn_gemm_nt_acc_ok(lda, lda, M, N, K)
    nparts = lib().ccn_stats_rows(M)
    dz_ref = torch.empty(M, N, device=DEV)
    sums_ref = torch.zeros((nparts + 1) * 2 * N, dtype=torch.float64, device=DEV)
    assert lib().ccn_gemm_nt(ptr(a), lda, ptr(w), lda, None, ptr(dz_ref), N, M, N, K, None, None) == 0
    assert lib().ccn_bn_act_bwd_reduce(ptr(dz_ref), N, ptr(y_prev), ldy, M, N, ptr(par[0]), ptr(par[1]), ptr(par[2]), ptr(par[3]),
                                       act, 0.01, ptr(sums_ref), None) == 0
    dz = torch.empty(M, N, device=DEV)
    sums = torch.zeros((nparts + 1) * 2 * N, dtype=torch.float64, device=DEV)
    rc = lib().ccn_gemm_nt_red(ptr(a), lda, ptr(w), lda, ptr(dz), N, M, N, K, ptr(y_prev), ldy, ptr(par), act, 0.01, ptr(sums), None)
    assert rc == 0, lib().ccn_last_error()
    torch.cuda.synchronize()
    assert torch.equal(dz, dz_ref)
    got, want = sums[: 2 * N], sums_ref[: 2 * N]
    scale = float(want.abs().max())
    assert float((got - want).abs().max()) <= 2e-6 * scale * (M ** 0.5), (float((got - want).abs().max()), scale)
    assert float((got - want).abs().max()) <= 1e-4 * scale
