"""ccn_gemm_nt_red against ccn_gemm_nt + ccn_bn_act_bwd_reduce, stand-alone at the KITTI step's deferred-layer shapes."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from curvecloudnet_amd._lib import lib, ptr  # noqa: E402

dev = "cuda"


def timeit(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    b, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    b.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return b.elapsed_time(e) / n


print("%-26s | product  + reduce  = separate | fused (ms)  | saved" % "M x N x K")
for m, n, k in [(1342781, 256, 256), (1342781, 192, 256), (1342781, 128, 192), (656150, 256, 256), (197729, 256, 512), (235102, 256, 256), (688586, 128, 128)]:
    a = torch.randn(m, k, device=dev); w = torch.randn(n, k, device=dev) * 0.05
    y = torch.randn(m, n, device=dev); dz = torch.empty(m, n, device=dev)
    par = (torch.rand(4, n, device=dev) + 0.5).contiguous()
    sums = torch.zeros((lib().ccn_stats_rows(m) + 1) * 2 * n, dtype=torch.float64, device=dev)
    t_p = timeit(lambda: lib().ccn_gemm_nt(ptr(a), k, ptr(w), k, None, ptr(dz), n, m, n, k, None, None))
    t_r = timeit(lambda: lib().ccn_bn_act_bwd_reduce(ptr(dz), n, ptr(y), n, m, n, ptr(par[0]), ptr(par[1]), ptr(par[2]), ptr(par[3]), 1, 0.01, ptr(sums), None))
    t_f = timeit(lambda: lib().ccn_gemm_nt_red(ptr(a), k, ptr(w), k, ptr(dz), n, m, n, k, ptr(y), n, ptr(par), 1, 0.01, ptr(sums), None))
    print("%9d x %4d x %4d | %6.3f  + %6.3f  = %6.3f   | %6.3f      | %5.1f %%" % (m, n, k, t_p, t_r, t_p + t_r, t_f, 100 * (1 - t_f / (t_p + t_r))))
