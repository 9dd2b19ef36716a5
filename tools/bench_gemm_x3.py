"""Split-bf16 ("bf16x3") product against the fp32 and plain-bf16 MFMA entries: error versus an fp64 product on a row
sample, and throughput at the GEMM shapes of the KITTI bench workload (profiles/archive/r01m_kitti_gemm_shapes.txt)."""
import sys
import torch
from curvecloudnet_amd._lib import call, ptr, lib
from curvecloudnet_amd.ops import _rows, _ld

SHAPES = [  # (rows M, N, K)
    (1342781, 256, 256), (58660, 1024, 1024), (197729, 512, 512), (656150, 256, 256), (1342781, 192, 256),
    (10550, 1024, 1024), (80365, 512, 512), (2341754, 64, 64), (688586, 128, 128), (1342781, 256, 192),
    (498380, 160, 262), (3168, 1024, 2048), (208234, 128, 128),
]
if len(sys.argv) > 1:
    SHAPES = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
dev = "cuda"
import os
if os.environ.get("X3_HOOK"):
    lib().ccn_gemm_x3_use_persistent(int(os.environ["X3_HOOK"]))


def timeit(fn, n=5):
    fn(); fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        b, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        b.record()
        for _ in range(n):
            fn()
        e.record(); torch.cuda.synchronize()
        best = min(best, b.elapsed_time(e) / n)
    return best


print("%-26s %27s   %27s" % ("M x K -> N", "TFLOP/s  fp32 / x3 / bf16", "max err vs fp64 / sum|a||w|"))
for m, n, k in SHAPES:
    torch.manual_seed(1)
    x = _rows(m, k, dev); x.normal_()
    w = _rows(n, k, dev, zero=True); w[:, :k].normal_(); w.mul_(0.05)
    bias = torch.randn(n, device=dev)
    stats = torch.empty((lib().ccn_stats_rows(m) + 1) * 2 * n, dtype=torch.float64, device=dev)
    nb = lib().ccn_gemm_x3_workspace_bytes(n, k)
    scratch = torch.empty(nb, dtype=torch.uint8, device=dev)
    ys = {}
    res = {}
    for name in ("gemm_nt", "gemm_nt_x3", "gemm_nt_bf16"):
        y = _rows(m, n, dev)
        extra = (ptr(scratch), nb) if name == "gemm_nt_x3" else ()
        fn = lambda: call(name, ptr(x), _ld(x), ptr(w), _ld(w), ptr(bias), ptr(y), _ld(y), m, n, k, ptr(stats), *extra)
        res[name] = timeit(fn)
        ys[name] = y
    rows = torch.randint(0, m, (512,), device=dev)
    xs, wd = x[rows, :k].double(), w[:, :k].double()
    ref = xs @ wd.t() + bias.double()
    scale = (xs.abs() @ wd.abs().t()) + bias.abs().double()
    errs = [float(((ys[nm][rows, :n].double() - ref).abs() / scale).max()) for nm in ("gemm_nt", "gemm_nt_x3", "gemm_nt_bf16")]
    fl = 2.0 * m * n * k
    print("%9d x %4d -> %4d  %8.1f %8.1f %8.1f   %9.2e %9.2e %9.2e   ms %.3f %.3f %.3f" % (
        (m, k, n) + tuple(fl / (res[nm] * 1e-3) / 1e12 for nm in ("gemm_nt", "gemm_nt_x3", "gemm_nt_bf16")) + tuple(errs)
        + tuple(res[nm] for nm in ("gemm_nt", "gemm_nt_x3", "gemm_nt_bf16"))), flush=True)
    del x, w, ys, stats
