"""One GEMM shape in a loop (for rocprofv3 --pmc): python tools/gemm_one_shape.py M K N [reps]"""
import sys
import torch
from curvecloudnet_amd._lib import call, ptr, lib
from curvecloudnet_amd.ops import _rows, _ld

m, k, n = (int(v) for v in sys.argv[1:4])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
dev = "cuda"
x = _rows(m, k, dev); x.normal_(); w = _rows(n, k, dev, zero=True); w.normal_(); w.mul_(0.05)
y = _rows(m, n, dev)
stats = torch.empty((lib().ccn_stats_rows(m) + 1) * 2 * n, dtype=torch.float64, device=dev)
for _ in range(reps):
    call("gemm_nt", ptr(x), _ld(x), ptr(w), _ld(w), None, ptr(y), _ld(y), m, n, k, ptr(stats))
torch.cuda.synchronize()
