"""A/B of the weight-gradient kernel's tile choice (ccn_gemm_tn_use_dma(2 / 3)) at the KITTI shapes whose N or K is not a multiple of 128."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from curvecloudnet_amd._lib import call, lib, ptr, workspace  # noqa: E402
from curvecloudnet_amd.ops import _ld, _rows  # noqa: E402

SHAPES = [(1342781, 256, 192), (1342781, 192, 128), (498380, 160, 262), (208234, 256, 259), (10550, 1024, 2051),
          (35151, 512, 1027), (78136, 256, 515), (1342781, 256, 256), (1342781, 128, 64), (688586, 64, 128)]
dev = "cuda"


def timeit(fn, n=6):
    fn(); fn(); torch.cuda.synchronize()
    b, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    b.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return b.elapsed_time(e) / n


print("%-28s %10s %10s %10s   (TFLOP/s: 64-wide over a dimension with a small remainder / 128 whenever > 64 / split)" % ("M x N x K", "64-rule", "128", "split"))
for m, n, k in SHAPES:
    dy = _rows(m, n, dev); dy.normal_(); x = _rows(m, k, dev); x.normal_()
    row, outs = [], []
    for mode in (2, 3, 4):
        lib().ccn_gemm_tn_use_dma(mode)
        nb = lib().ccn_gemm_tn_workspace_bytes(m, n, k)
        ws = workspace(nb, dev)
        dw = _rows(n, k, dev, zero=True)
        t = min(timeit(lambda: call("gemm_tn_ws", ptr(dy), _ld(dy), ptr(x), _ld(x), ptr(dw), _ld(dw), m, n, k, ptr(ws), nb)) for _ in range(2))
        dw.zero_()
        call("gemm_tn_ws", ptr(dy), _ld(dy), ptr(x), _ld(x), ptr(dw), _ld(dw), m, n, k, ptr(ws), nb)
        outs.append(dw.clone())
        row.append("%10.1f" % (2.0 * m * n * k / 1e9 / t))
    lib().ccn_gemm_tn_use_dma(4)
    rel = float(max((outs[0] - o).abs().max() for o in outs[1:]) / outs[0].abs().max())
    print("%9d x %4d x %4d   %s   max rel diff %.1e" % (m, n, k, " ".join(row), rel))
    del dy, x, dw, ws, outs
