"""One GEMM shape in a loop (for rocprofv3 --pmc): python tools/gemm_one_shape.py M K N [reps] [entry]
(entry: gemm_nt (default), gemm_nt_x3, gemm_nt_bf16)"""
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
entry = sys.argv[5] if len(sys.argv) > 5 else "gemm_nt"
extra = ()
if entry == "gemm_nt_x3":
    nb = lib().ccn_gemm_x3_workspace_bytes(n, k)
    scratch = torch.empty(nb, dtype=torch.uint8, device=dev)
    extra = (ptr(scratch), nb)
for _ in range(reps):
    call(entry, ptr(x), _ld(x), ptr(w), _ld(w), None, ptr(y), _ld(y), m, n, k, ptr(stats), *extra)
torch.cuda.synchronize()
