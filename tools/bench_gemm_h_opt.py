"""ccn_gemm_nt_h ablations (timing only): python tools/bench_gemm_h_opt.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from curvecloudnet_amd._lib import call, lib, ptr  # noqa: E402

dev = "cuda"
OPTS = [int(v, 0) for v in sys.argv[1:]] or [0, 1, 2, 3]


def timeit(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    b, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    b.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return b.elapsed_time(e) / n


print("%-24s | ms at opt = 0 (all), 1 (no stores), 2 (no wait for copies), 3 (neither)   [fp32 result | 16-bit result]" % "M x N x K")
for m, n, k in [(1870000, 256, 256), (1870000, 128, 64), (913000, 256, 256), (275000, 512, 512), (81000, 1024, 1024)]:
    x16 = torch.randn(m, k, device=dev).to(torch.bfloat16); w16 = (torch.randn(n, k, device=dev) * 0.05).to(torch.bfloat16)
    y = torch.empty(m, n, device=dev); y16 = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
    row = []
    for out16 in (0, 1):
        for opt in OPTS:
            lib().ccn_gemm_h_opt(opt)
            row.append(timeit(lambda: call("gemm_nt_h", ptr(x16), k, ptr(w16), k, None, ptr(y16 if out16 else y), n, m, n, k, None, 0, out16)))
    lib().ccn_gemm_h_opt(0)
    half = len(row) // 2
    print("%9d x %4d x %4d | %s | %s" % (m, n, k, " ".join("%6.3f" % v for v in row[:half]), " ".join("%6.3f" % v for v in row[half:])))
    del x16, w16, y, y16
