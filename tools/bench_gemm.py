"""GPU microbenchmark of the three GEMM entry points at the shapes of the bench workload."""
import sys
import torch
from curvecloudnet_amd._lib import call, ptr, lib
from curvecloudnet_amd.ops import _rows, _ld

SHAPES = [  # (rows, C_in, C_out)
    (4200000, 134, 64), (4200000, 64, 64), (4200000, 262, 128), (4200000, 128, 128),
    (1200000, 38, 64), (1200000, 64, 128), (1200000, 128, 192), (1200000, 192, 256), (1200000, 256, 256),
    (200000, 259, 256), (500000, 1310, 32), (400000, 160, 128), (400000, 64, 20),
]
which = sys.argv[1:] or ["nt", "nn", "tn"]
dev = "cuda"
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    b, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    b.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return b.elapsed_time(e) / n
print("%-26s %10s %10s %10s   (TFLOP/s)" % ("rows x Cin -> Cout", *("%s" % w for w in ["nt", "nn", "tn"])))
for m, k, n in SHAPES:
    x = _rows(m, k, dev); x.normal_(); w = _rows(n, k, dev, zero=True); w.normal_(); w.mul_(0.05)
    y = _rows(m, n, dev); y.normal_(); dx = _rows(m, k, dev); dw = _rows(n, k, dev, zero=True)
    stats = torch.empty((lib().ccn_stats_rows(m) + 1) * 2 * n, dtype=torch.float64, device=dev)
    res = {}
    if "nt" in which:
        res["nt"] = timeit(lambda: call("gemm_nt", ptr(x), _ld(x), ptr(w), _ld(w), None, ptr(y), _ld(y), m, n, k, ptr(stats)))
    if "nn" in which:
        res["nn"] = timeit(lambda: call("gemm_nn", ptr(y), _ld(y), ptr(w), _ld(w), ptr(dx), _ld(dx), m, n, k))
    if "tn" in which:
        res["tn"] = timeit(lambda: call("gemm_tn", ptr(y), _ld(y), ptr(x), _ld(x), ptr(dw), _ld(dw), m, n, k))
    fl = 2.0 * m * n * k
    print("%9d x %4d -> %4d " % (m, k, n) + " ".join("%10.1f" % (fl / (res[w] * 1e-3) / 1e12) if w in res else "%10s" % "-" for w in ["nt", "nn", "tn"])
          + "   ms: " + " ".join("%.3f" % res[w] for w in res))
    del x, w, y, dx, dw, stats
