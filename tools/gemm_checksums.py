"""Bit patterns of ccn_gemm_nt / _acc results on seeded operands, as checksums: run once per library build (CCN_LIB_PATH) and
diff the outputs -- a kernel change that claims "the same bits" prints the same lines.
    python tools/gemm_checksums.py > a.txt;  CCN_LIB_PATH=... python tools/gemm_checksums.py > b.txt;  diff a.txt b.txt"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from curvecloudnet_amd._lib import call, lib, ptr  # noqa: E402
from curvecloudnet_amd.ops import _ld, _rows  # noqa: E402

SHAPES = [(70001, 256, 256), (66000, 512, 512), (20000, 1024, 1024), (131073, 128, 96), (40000, 192, 128), (30000, 259, 262),
          (100000, 256, 64), (50000, 300, 515)]
dev = "cuda"
for m, n, k in SHAPES:
    g = torch.Generator().manual_seed(m + n + k)
    x = _rows(m, k, dev); x[:, :k].copy_(torch.randn(m, k, generator=g).to(dev))
    w = _rows(n, k, dev, zero=True); w[:, :k].copy_((torch.randn(n, k, generator=g) * 0.05).to(dev))
    b = torch.randn(n, generator=g).to(dev)
    nparts = lib().ccn_stats_rows(m)
    for stats in (True, False):
        y = _rows(m, n, dev); y.zero_()
        st = torch.zeros((nparts + 1) * 2 * n, dtype=torch.float64, device=dev)
        call("gemm_nt", ptr(x), _ld(x), ptr(w), _ld(w), ptr(b), ptr(y), _ld(y), m, n, k, ptr(st) if stats else None)
        bits = y[:, :n].contiguous().view(torch.int32).to(torch.int64)
        print("gemm_nt %7d x %4d x %4d stats=%d  sum(bits)=%d  xor-fold=%d  stats-sum=%.17g" % (
            m, n, k, stats, int(bits.sum()), int((bits * torch.arange(1, bits.numel() + 1, device=dev).view_as(bits) % 1000003).sum()),
            float(st[: nparts * 2 * n].sum())))
    if lib().ccn_gemm_nt_acc_ok(_ld(x), _ld(w), m, n, k):
        y = _rows(m, n, dev); y[:, :n].copy_(torch.randn(m, n, generator=g).to(dev))
        call("gemm_nt_acc", ptr(x), _ld(x), ptr(w), _ld(w), ptr(y), _ld(y), m, n, k)
        bits = y[:, :n].contiguous().view(torch.int32).to(torch.int64)
        print("gemm_nt_acc %7d x %4d x %4d  sum(bits)=%d" % (m, n, k, int(bits.sum())))
