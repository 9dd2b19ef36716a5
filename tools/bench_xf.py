"""ccn_gemm_nt_xf (BatchNorm + activation of the previous layer applied to the A fragments) against ccn_bn_act_fwd + ccn_gemm_nt:
same bits, and what the two ways cost."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from curvecloudnet_amd._lib import call, lib, ptr  # noqa: E402
from curvecloudnet_amd.ops import _ld, _rows  # noqa: E402

SHAPES = [(1342781, 192, 128), (1342781, 256, 192), (1342781, 128, 64), (688586, 128, 128), (235102, 256, 256), (197729, 512, 512),
          (58660, 1024, 1024), (208234, 128, 256)]
dev = "cuda"


def timeit(fn, n=6):
    fn(); fn(); torch.cuda.synchronize()
    b, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    b.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return b.elapsed_time(e) / n


print("%-26s %10s %10s %10s %10s   (ms: bn_act_fwd, gemm_nt, their sum, gemm_nt_xf)" % ("M x N x K", "bn_act", "gemm", "sum", "xf"))
for m, n, k in SHAPES:
    y0 = _rows(m, k, dev); y0.normal_()
    w = _rows(n, k, dev); w.normal_(); w.mul_(k ** -0.5)
    par = torch.empty(2, k, device=dev); par[0].uniform_(0.5, 1.5); par[1].normal_()
    z = _rows(m, k, dev); out1 = _rows(m, n, dev); out2 = _rows(m, n, dev)
    stats = torch.empty((lib().ccn_stats_rows(m) + 1) * 2 * n, dtype=torch.float64, device=dev)
    ok = lib().ccn_gemm_nt_xf_ok(_ld(y0), _ld(w), m, n, k)
    for act in (1, 2):
        t_bn = timeit(lambda: call("bn_act_fwd", ptr(y0), _ld(y0), m, k, ptr(par[0]), ptr(par[1]), act, 0.01, ptr(z), _ld(z)))
        t_g = timeit(lambda: call("gemm_nt", ptr(z), _ld(z), ptr(w), _ld(w), None, ptr(out1), _ld(out1), m, n, k, ptr(stats)))
        s1 = stats.clone()
        if not ok:
            print("%9d x %4d x %4d  act %d: not eligible" % (m, n, k, act)); continue
        t_x = timeit(lambda: call("gemm_nt_xf", ptr(y0), _ld(y0), ptr(par[0]), ptr(par[1]), act, 0.01, ptr(w), _ld(w), None,
                                  ptr(out2), _ld(out2), m, n, k, ptr(stats)))
        np_ = lib().ccn_stats_rows(m) * 2 * n
        same = torch.equal(out1, out2) and torch.equal(s1[:np_], stats[:np_])
        print("%9d x %4d x %4d  act %d %10.3f %10.3f %10.3f %10.3f   %s" % (m, n, k, act, t_bn, t_g, t_bn + t_g, t_x,
                                                                             "same bits" if same else "DIFFERENT"))
    del y0, w, z, out1, out2, stats

print()
print("%-26s %10s %10s   (ms: gemm_tn_ws on the stored activation, gemm_tn_ws_xf on the pre-normalisation product)" % ("M x N x K", "tn", "tn_xf"))
from curvecloudnet_amd._lib import workspace  # noqa: E402
for m, n, k in [(1342781, 192, 128), (1342781, 256, 192), (688586, 128, 128), (235102, 256, 256), (197729, 512, 512),
                (58660, 1024, 1024), (208234, 256, 259)]:
    y0 = _rows(m, k, dev); y0.normal_()
    dy = _rows(m, n, dev); dy.normal_()
    par = torch.empty(2, k, device=dev); par[0].uniform_(0.5, 1.5); par[1].normal_()
    z = _rows(m, k, dev)
    nb = lib().ccn_gemm_tn_workspace_bytes(m, n, k)
    ws = workspace(nb, dev)
    for act in (1, 2):
        call("bn_act_fwd", ptr(y0), _ld(y0), m, k, ptr(par[0]), ptr(par[1]), act, 0.01, ptr(z), _ld(z))
        d1, d2 = _rows(n, k, dev, zero=True), _rows(n, k, dev, zero=True)
        t1 = timeit(lambda: call("gemm_tn_ws", ptr(dy), _ld(dy), ptr(z), _ld(z), ptr(d1), _ld(d1), m, n, k, ptr(ws), nb))
        t2 = timeit(lambda: call("gemm_tn_ws_xf", ptr(dy), _ld(dy), ptr(y0), _ld(y0), ptr(par[0]), ptr(par[1]), act, 0.01, ptr(d2),
                                 _ld(d2), m, n, k, ptr(ws), nb))
        print("%9d x %4d x %4d  act %d %10.3f %10.3f   %s" % (m, n, k, act, t1, t2, "same bits" if torch.equal(d1, d2) else
              "DIFFERENT (max rel %.2e)" % float((d1 - d2).abs().max() / d1.abs().max())))
    del y0, dy, z, ws
