"""The 16-bit storage GEMMs (ccn_gemm_nt_h / ccn_gemm_tn_h) against the register-staged bf16 kernels on fp32 rows
(ccn_gemm_nt_bf16 / ccn_gemm_tn_bf16), stand-alone, at the shapes of BASELINE configs[2]: TFLOP/s and algorithmic GB/s."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from curvecloudnet_amd._lib import call, lib, ptr, workspace  # noqa: E402

dev = "cuda"
SHAPES = [(1870000, 256, 256), (1870000, 192, 128), (1870000, 128, 64), (913000, 256, 256), (557000, 64, 64),
          (275000, 512, 512), (81000, 1024, 1024), (14700, 1024, 1024), (290000, 128, 128)]


def timeit(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    b, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    b.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return b.elapsed_time(e) / n


print("%-24s | NT: fp32 rows (old)   16-bit rows (new: TFLOP/s, GB/s)  16-bit result | TN: old      new (TFLOP/s, GB/s)" % "M x N x K")
for m, n, k in SHAPES:
    x32 = torch.randn(m, k, device=dev); w32 = torch.randn(n, k, device=dev) * 0.05
    x16 = x32.to(torch.bfloat16); w16 = w32.to(torch.bfloat16)
    y = torch.empty(m, n, device=dev); y16 = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
    stats = torch.empty((lib().ccn_stats_rows(m) + 1) * 2 * n, dtype=torch.float64, device=dev)
    fl = 2.0 * m * n * k / 1e9
    t_old = timeit(lambda: call("gemm_nt_bf16", ptr(x32), k, ptr(w32), k, None, ptr(y), n, m, n, k, ptr(stats)))
    t_new = timeit(lambda: call("gemm_nt_h", ptr(x16), k, ptr(w16), k, None, ptr(y), n, m, n, k, ptr(stats), 0, 0))
    t_16 = timeit(lambda: call("gemm_nt_h", ptr(x16), k, ptr(w16), k, None, ptr(y16), n, m, n, k, None, 0, 1))
    dy32 = torch.randn(m, n, device=dev); dy16 = dy32.to(torch.bfloat16)
    dw = torch.zeros(n, k, device=dev)
    nb = lib().ccn_gemm_tn_h_workspace_bytes(m, n, k); ws = workspace(nb, dev)
    t_told = timeit(lambda: call("gemm_tn_bf16", ptr(dy32), n, ptr(x32), k, ptr(dw), k, m, n, k))
    t_tnew = timeit(lambda: call("gemm_tn_h", ptr(dy16), n, ptr(x16), k, ptr(dw), k, m, n, k, ptr(ws), nb))
    print("%9d x %4d x %4d | %6.1f   %6.1f %6.0f   %6.1f %6.0f | %6.1f   %6.1f %6.0f"
          % (m, n, k, fl / t_old, fl / t_new, m * (2 * k + 4 * n) / t_new / 1e6, fl / t_16, m * (2 * k + 2 * n) / t_16 / 1e6,
             fl / t_told, fl / t_tnew, m * (n + k) * 2 / t_tnew / 1e6))
    del x32, w32, x16, w16, y, y16, dy32, dy16

print("\nelementwise 16-bit kernels, GB/s (algorithmic bytes / time)")
print("%-16s | cast    bn_fwd_h  reduce_h  apply_h(dz16)  apply_h(dz32)" % "rows x C")
for m, c in [(1870000, 256), (3227000, 64), (818000, 128), (198000, 512), (59000, 1024), (290000, 259)]:
    x = torch.randn(m, c, device=dev)
    ld16 = (c + 7) // 8 * 8
    o16 = torch.empty(m, ld16, dtype=torch.bfloat16, device=dev)
    g16 = torch.randn(m, ld16, device=dev).to(torch.bfloat16)
    par = torch.rand(4, c, device=dev) + 0.5
    sums = torch.zeros((lib().ccn_stats_rows(m) + 1) * 2 * c, dtype=torch.float64, device=dev)
    dgb = torch.zeros(2, c, device=dev)
    t_c = timeit(lambda: call("cast_rows_h", ptr(x), c, m, c, ptr(o16), ld16, 0))
    t_f = timeit(lambda: call("bn_act_fwd_h", ptr(x), c, m, c, ptr(par[0]), ptr(par[1]), 1, 0.2, ptr(o16), ld16, 0))
    t_r = timeit(lambda: call("bn_act_bwd_reduce_h", ptr(g16), ld16, ptr(x), c, m, c, ptr(par[0]), ptr(par[1]), ptr(par[2]),
                              ptr(par[3]), 1, 0.2, ptr(sums)))
    t_a = timeit(lambda: call("bn_act_bwd_apply_h", ptr(g16), 1, ld16, ptr(x), c, m, c, ptr(par[0]), ptr(par[1]), ptr(par[2]),
                              ptr(par[3]), 1, 0.2, ptr(sums), float(m), 1, 0, ptr(o16), ld16, ptr(dgb[0]), ptr(dgb[1]), 0))
    g32 = torch.randn(m, c, device=dev)
    t_a32 = timeit(lambda: call("bn_act_bwd_apply_h", ptr(g32), 0, c, ptr(x), c, m, c, ptr(par[0]), ptr(par[1]), ptr(par[2]),
                                ptr(par[3]), 1, 0.2, ptr(sums), float(m), 1, 0, ptr(o16), ld16, ptr(dgb[0]), ptr(dgb[1]), 0))
    e = m * c / 1e6
    print("%8d x %4d | %6.0f  %6.0f   %6.0f   %6.0f   %6.0f" % (m, c, 6 * e / t_c, 6 * e / t_f, 6 * e / t_r, 8 * e / t_a, 10 * e / t_a32))
    del x, o16, g16, g32
