"""GPU microbenchmark of gemm_nt (with / without BatchNorm statistics) and gemm_tn at the dominant KITTI-bench shapes."""
import sys
import torch
from curvecloudnet_amd._lib import call, ptr, lib
from curvecloudnet_amd.ops import _rows, _ld

SHAPES = [(1342781, 64, 128), (2341754, 64, 64), (1342781, 96, 128), (1342781, 256, 256), (67368, 1024, 1024), (224448, 512, 512), (4408488, 64, 64), (1652112, 128, 128),
          (10550, 1024, 1024), (35151, 512, 512)]
dev = "cuda"


def timeit(fn, n=6):
    fn(); fn(); torch.cuda.synchronize()
    b, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    b.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return b.elapsed_time(e) / n


name = sys.argv[1] if len(sys.argv) > 1 else "gemm_nt"
if len(sys.argv) > 2:
    lib().ccn_gemm_use_dma(int(sys.argv[2]))     # 0: register-staged kernels only, 2: DMA without the persistent loop
print("%-28s %12s %12s %12s   (TFLOP/s)" % ("M x K -> N", "nt+stats", "nt", "tn"))
for m, k, n in SHAPES:
    x = _rows(m, k, dev); x.normal_(); w = _rows(n, k, dev, zero=True); w.normal_(); w.mul_(0.05)
    y = _rows(m, n, dev); y.normal_(); dw = _rows(n, k, dev, zero=True)
    stats = torch.empty((lib().ccn_stats_rows(m) + 1) * 2 * n, dtype=torch.float64, device=dev)
    f_stats = lambda: call(name, ptr(x), _ld(x), ptr(w), _ld(w), None, ptr(y), _ld(y), m, n, k, ptr(stats))   # noqa: E731
    f_plain = lambda: call(name, ptr(x), _ld(x), ptr(w), _ld(w), None, ptr(y), _ld(y), m, n, k, None)         # noqa: E731
    t2 = timeit(f_plain)
    t1 = timeit(f_stats)
    t2 = min(t2, timeit(f_plain))
    t1 = min(t1, timeit(f_stats))
    t3 = timeit(lambda: call("gemm_tn", ptr(y), _ld(y), ptr(x), _ld(x), ptr(dw), _ld(dw), m, n, k))
    fl = 2.0 * m * n * k / 1e9
    print("%9d x %4d -> %4d  %12.1f %12.1f %12.1f   ms: %.3f %.3f %.3f" % (m, k, n, fl / t1, fl / t2, fl / t3, t1, t2, t3))
    del x, w, y, dw, stats
