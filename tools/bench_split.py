"""Tail split of the paired fp32 kernel (ccn_gemm_nt_ws) against the unsplit product at the KITTI step's few-tile shapes
(profiles/r04_kitti_gemm_shapes.txt: 3-7 tiles per workgroup slot): ms and TFLOP/s, parts used."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from curvecloudnet_amd._lib import call, lib, ptr  # noqa: E402
from curvecloudnet_amd.ops import _ld, _rows  # noqa: E402

SHAPES = [(10550, 1024, 1024), (35151, 512, 512), (78136, 256, 256), (10550, 512, 1024), (10550, 1024, 512), (35151, 256, 512),
          (3168, 2048, 3072), (3168, 3072, 2048), (3168, 1024, 2048), (10550, 1024, 2051), (10550, 2051, 1024), (35151, 512, 1027),
          (35151, 1027, 512), (80365, 512, 512), (58660, 1024, 1024), (57232, 1024, 1024), (197729, 512, 512), (235102, 256, 256)]
dev = "cuda"


def timeit(fn, n=8):
    fn(); fn(); torch.cuda.synchronize()
    b, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    b.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return b.elapsed_time(e) / n


nb = int(lib().ccn_gemm_nt_split_workspace_bytes())
ws = torch.zeros(nb, dtype=torch.uint8, device=dev)
print("%-24s %6s %6s %9s %9s %8s %8s" % ("M x N x K", "tiles", "parts", "plain ms", "split ms", "TF/s", "TF/s"))
for m, n, k in SHAPES:
    x = _rows(m, k, dev); x.normal_()
    w = _rows(n, k, dev, zero=True); w.normal_(); w.mul_(k ** -0.5)
    y = _rows(m, n, dev)
    parts = lib().ccn_gemm_nt_split_parts(m, n, k, nb)
    t0 = timeit(lambda: call("gemm_nt", ptr(x), _ld(x), ptr(w), _ld(w), None, ptr(y), _ld(y), m, n, k, None))
    t1 = timeit(lambda: call("gemm_nt_ws", ptr(x), _ld(x), ptr(w), _ld(w), None, ptr(y), _ld(y), m, n, k, None, ptr(ws), nb))
    fl = 2.0 * m * n * k / 1e9
    print("%8d x %4d x %4d %6d %6d %9.3f %9.3f %8.1f %8.1f" % (m, n, k, ((m + 127) // 128) * ((n + 127) // 128), parts, t0, t1,
                                                             fl / t0, fl / t1))
    del x, w, y
