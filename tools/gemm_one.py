"""One GEMM shape per entry point (for rocprofv3 --pmc runs): 4.2M x 128 -> 128 and 4.2M x 262 -> 128."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from curvecloudnet_amd._lib import call, ptr, lib
from curvecloudnet_amd.ops import _rows, _ld
dev = "cuda"
for m, k, n in ((4200000, 128, 128), (4200000, 262, 128), (4200000, 64, 64)):
    x = _rows(m, k, dev); x.normal_(); w = _rows(n, k, dev, zero=True); w.normal_()
    y = _rows(m, n, dev); y.normal_(); dx = _rows(m, k, dev); dw = _rows(n, k, dev, zero=True)
    stats = torch.empty((lib().ccn_stats_rows(m) + 1) * 2 * n, dtype=torch.float64, device=dev)
    for _ in range(2):
        call("gemm_nt", ptr(x), _ld(x), ptr(w), _ld(w), None, ptr(y), _ld(y), m, n, k, ptr(stats))
        call("gemm_nn", ptr(y), _ld(y), ptr(w), _ld(w), ptr(dx), _ld(dx), m, n, k)
        call("gemm_tn", ptr(y), _ld(y), ptr(x), _ld(x), ptr(dw), _ld(dw), m, n, k)
    torch.cuda.synchronize()
