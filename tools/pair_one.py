"""One shape of the paired GEMM kernel, a few launches (for rocprofv3 --pmc runs): python tools/pair_one.py M N K"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from curvecloudnet_amd._lib import call, ptr  # noqa: E402
from curvecloudnet_amd.ops import _ld, _rows  # noqa: E402

m, n, k = (int(a) for a in sys.argv[1:4])
x = _rows(m, k, "cuda"); x.normal_(); w = _rows(n, k, "cuda"); w.normal_(); y = _rows(m, n, "cuda")
for _ in range(6):
    call("gemm_nt", ptr(x), _ld(x), ptr(w), _ld(w), None, ptr(y), _ld(y), m, n, k, None)
torch.cuda.synchronize()
