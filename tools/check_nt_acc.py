"""ccn_gemm_nt_acc (Y += A W^T, the paired kernel's accumulate variant) against fp64, and its rate next to the plain product."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from curvecloudnet_amd._lib import call, lib, ptr  # noqa: E402

dev = "cuda"


def timeit(fn, n=8):
    fn(); fn(); torch.cuda.synchronize()
    b, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    b.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return b.elapsed_time(e) / n


for m, n, k in [(1342781, 256, 256), (197729, 512, 256), (20000, 256, 128), (16500, 200, 64), (70001, 132, 96)]:
    assert lib().ccn_gemm_nt_acc_ok(k, k, m, n, k)
    gen = torch.Generator().manual_seed(m)
    x = torch.randn(m, k, generator=gen).to(dev); w = (torch.randn(n, k, generator=gen) * 0.1).to(dev)
    y0 = torch.randn(m, n, generator=gen).to(dev)
    y = y0.clone()
    call("gemm_nt_acc", ptr(x), k, ptr(w), k, ptr(y), n, m, n, k)
    rows = torch.cat([torch.arange(0, 300), torch.arange(m - 300, m), torch.randint(0, m, (400,))]).to(dev)
    ref = y0[rows].double() + x[rows].double() @ w.double().t()
    err = float((y[rows].double() - ref).abs().max() / ref.abs().max())
    t_acc = timeit(lambda: call("gemm_nt_acc", ptr(x), k, ptr(w), k, ptr(y), n, m, n, k))
    t_nt = timeit(lambda: call("gemm_nt", ptr(x), k, ptr(w), k, None, ptr(y), n, m, n, k, None))
    fl = 2.0 * m * n * k / 1e9
    print("%8d x %4d x %4d  max rel err %.2e   acc %.1f TFLOP/s   plain %.1f" % (m, n, k, err, fl / t_acc, fl / t_nt))
    assert err < 5e-6, err
