"""ccn_gemm_nt_h with an fp32 result: 128 x 64 tiles / three workgroups per CU (ccn_gemm_h_opt bit 5 = always) against 128 x 128 /
two (bit 4 = never; the default picks per shape), stand-alone at the shapes of BASELINE configs[2] / [4]: TFLOP/s and algorithmic GB/s."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from curvecloudnet_amd._lib import call, lib, ptr  # noqa: E402

dev = "cuda"
SHAPES = [(1870000, 256, 256), (1870000, 192, 128), (1870000, 128, 64), (913000, 256, 256), (557000, 64, 64),
          (275000, 512, 512), (81000, 1024, 1024), (14700, 1024, 1024), (290000, 128, 128), (1870000, 256, 192),
          (913000, 259, 256), (290000, 2051, 1024)]


def timeit(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    b, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    b.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return b.elapsed_time(e) / n


print("%-26s | 128x128 tiles: TFLOP/s  GB/s | 128x64 tiles: TFLOP/s  GB/s | ratio" % "M x N x K (stats on)")
for m, n, k in SHAPES:
    kp = (k + 7) // 8 * 8
    x16 = torch.randn(m, kp, device=dev).to(torch.bfloat16); w16 = (torch.randn(n, kp, device=dev) * 0.05).to(torch.bfloat16)
    y = torch.empty(m, n, device=dev)
    stats = torch.empty((lib().ccn_stats_rows(m) + 1) * 2 * n, dtype=torch.float64, device=dev)
    fl, by = 2.0 * m * n * k / 1e9, m * (2.0 * k + 4.0 * n) / 1e6
    res = []
    for opt in (16, 32):
        lib().ccn_gemm_h_opt(opt)
        t = timeit(lambda: call("gemm_nt_h", ptr(x16), kp, ptr(w16), kp, None, ptr(y), n, m, n, k, ptr(stats), 0, 0))
        res.append(t)
    lib().ccn_gemm_h_opt(0)
    print("%9d x %4d x %4d | %8.1f %8.0f | %8.1f %8.0f | %.2f" % (m, n, k, fl / res[0], by / res[0], fl / res[1], by / res[1], res[0] / res[1]))
    del x16, w16, y
