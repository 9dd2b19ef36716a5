"""A/B of the paired GEMM kernel's micro-options (ccn_gemm_pair_opt) at the dominant KITTI shapes, stand-alone."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from curvecloudnet_amd._lib import call, lib, ptr  # noqa: E402
from curvecloudnet_amd.ops import _ld, _rows  # noqa: E402

SHAPES = [(1342781, 256, 256), (656150, 256, 256), (197729, 512, 512), (58660, 1024, 1024), (208234, 128, 128),
          (1342781, 192, 256), (10550, 1024, 1024), (208234, 256, 259), (1342781, 192, 128), (498380, 262, 160),
          (208234, 259, 256), (498380, 160, 262)]
if os.environ.get("SHAPES") == "narrow":      # (the 8-wave persistent kernel's shapes: N <= 64)
    SHAPES = [(2341754, 64, 64), (2134741, 64, 64), (688586, 64, 128), (617950, 64, 64)]
if os.environ.get("SHAPES") == "odd":
    SHAPES = [sh for sh in SHAPES if sh[1] % 128]
dev = "cuda"


def timeit(fn, n=8):
    fn(); fn(); torch.cuda.synchronize()
    b, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    b.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return b.elapsed_time(e) / n


OPTS = [int(a, 0) for a in sys.argv[1:]] or list(range(4))
print("%-26s %s   (TFLOP/s, with BN statistics / without)" % ("M x N x K", "  ".join("opt=%-7d" % o for o in OPTS)))
for m, n, k in SHAPES:
    x = _rows(m, k, dev); x.normal_(); w = _rows(n, k, dev, zero=True); w.normal_(); w.mul_(0.05)
    y = _rows(m, n, dev)
    stats = torch.empty((lib().ccn_stats_rows(m) + 1) * 2 * n, dtype=torch.float64, device=dev)
    ref = None
    row = []
    for opt in OPTS:
        lib().ccn_gemm_pair_opt(opt)
        t1 = min(timeit(lambda: call("gemm_nt", ptr(x), _ld(x), ptr(w), _ld(w), None, ptr(y), _ld(y), m, n, k, ptr(stats)))
                 for _ in range(2))
        t2 = min(timeit(lambda: call("gemm_nt", ptr(x), _ld(x), ptr(w), _ld(w), None, ptr(y), _ld(y), m, n, k, None))
                 for _ in range(2))
        if ref is None:
            ref = y.clone()
        assert torch.equal(ref, y) or (opt & 64) or 64 in [o & 64 for o in OPTS], "opt %d changes the result" % opt
        fl = 2.0 * m * n * k / 1e9
        row.append("%5.1f / %5.1f" % (fl / t1, fl / t2))
    lib().ccn_gemm_pair_opt(0)
    print("%9d x %4d x %4d   %s" % (m, n, k, "  ".join(row)))
    del x, w, y, stats
