"""Where the software-pipelined split-bf16 kernel's cycles go: the stamped diagnostic build (ccn_gemm_x3_debug) on one shape.
python tools/x3_stamps.py M N K   -- shares per wave (median over waves), in-kernel clock, matrix-pipe share."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from curvecloudnet_amd._lib import call, lib, ptr  # noqa: E402
from curvecloudnet_amd.ops import _ld, _rows  # noqa: E402

m, n, k = (int(a) for a in sys.argv[1:4])
dev = "cuda"
x = _rows(m, k, dev); x.normal_(); w = _rows(n, k, dev, zero=True); w.normal_(); w.mul_(0.05); y = _rows(m, n, dev)
stats = torch.empty((lib().ccn_stats_rows(m) + 1) * 2 * n, dtype=torch.float64, device=dev)
nb = lib().ccn_gemm_x3_workspace_bytes(n, k)
scratch = torch.empty(nb, dtype=torch.uint8, device=dev)
dbg = torch.zeros(512 * 4 * 16, dtype=torch.int64, device=dev)


def launch():
    call("gemm_nt_x3", ptr(x), _ld(x), ptr(w), _ld(w), None, ptr(y), _ld(y), m, n, k, ptr(stats), ptr(scratch), nb)


def run(stamped, reps=20):
    lib().ccn_gemm_x3_debug(ptr(dbg) if stamped else None)
    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    b, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    b.record()
    for _ in range(reps):
        launch()
    e.record(); torch.cuda.synchronize()
    lib().ccn_gemm_x3_debug(None)
    return b.elapsed_time(e) / reps


run(False, 300)      # warm the chip up first (clock under load)
t_plain = run(False)
t_st = run(True)
fl = 2.0 * m * n * k / 1e9
print("%d x %d x %d: plain %.3f ms = %.1f TFLOP/s, stamped build %.3f ms = %.1f TFLOP/s (incl. the weight split)"
      % (m, n, k, t_plain, fl / t_plain, t_st, fl / t_st))
d = dbg.view(512, 4, 16).cpu().double()
live = d[:, :, 7] > 0
tot, real = d[..., 0][live], d[..., 1][live]
clk = tot / real * 100e6 / 1e9
print("waves %d, lifetime cycles median %.0f (min %.0f max %.0f), in-kernel clock median %.3f GHz (min %.3f max %.3f)"
      % (int(live.sum()), tot.median(), tot.min(), tot.max(), clk.median(), clk.min(), clk.max()))
names = ["wait vmcnt", "barrier", "issue+readout", "MFMAs + next step's reads/split", "epilogue"]
nst = d[..., 7][live]
for i, nm in enumerate(names):
    v = d[..., 2 + i][live]
    print("  %-32s %5.1f %% of lifetime   (%.0f cycles per 16-deep step, median over waves; min %.0f max %.0f)"
          % (nm, float((v / tot).median()) * 100, float((v / nst).median()), float((v / nst).min()), float((v / nst).max())))
acc = sum(d[..., 2 + i][live] for i in range(5))
print("  %-32s %5.1f %%" % ("(unaccounted)", float(((tot - acc) / tot).median()) * 100))
mf = nst * 24 * 32
print("  MFMA cycles of the wave itself: %.1f %% of its lifetime (x2 waves per SIMD)" % (float((mf / tot).median()) * 100))
r0, r1 = d[..., 10][live], d[..., 11][live]
span = float(r1.max() - r0.min())
alive = float((r1 - r0).sum()) / float(live.sum())
print("  real-time span of all waves: %.3f ms; start spread %.1f us; end spread %.1f us; mean wave lifetime %.3f ms = %.1f %% "
      "of the span; workgroups %d" % (span / 1e5, (r0.max() - r0.min()) / 100, (r1.max() - r1.min()) / 100, alive / 1e5,
                                       100 * alive / span, int(live.sum()) // 4))
