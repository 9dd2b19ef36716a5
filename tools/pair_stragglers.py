"""Who finishes late in the paired fp32 GEMM kernel (512 persistent workgroups, equal tile counts): end times by XCD, by
CU, and by age within a CU's pair, from the stamped diagnostic build.  python tools/pair_stragglers.py M N K"""
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
dbg = torch.zeros(512 * 4 * 16, dtype=torch.int64, device=dev)
for _ in range(200):
    call("gemm_nt", ptr(x), _ld(x), ptr(w), _ld(w), None, ptr(y), _ld(y), m, n, k, None)
torch.cuda.synchronize()
lib().ccn_gemm_pair_debug(ptr(dbg))
for _ in range(5):
    call("gemm_nt", ptr(x), _ld(x), ptr(w), _ld(w), None, ptr(y), _ld(y), m, n, k, None)
torch.cuda.synchronize()
lib().ccn_gemm_pair_debug(None)
d = dbg.view(512, 4, 16).cpu()
r0, r1 = d[:, 0, 10].double(), d[:, 0, 11].double()          # wave 0 of every workgroup, 100 MHz ticks
t0 = float(r0.min())
start, end = (r0 - t0) / 100.0, (r1 - t0) / 100.0            # us
xcc = d[:, 0, 12] & 0xf
hw = d[:, 0, 13]
cu = (hw >> 8) & 0xf
sh = (hw >> 12) & 0x1
se = (hw >> 13) & 0x7
cuid = xcc * 1000 + se * 100 + sh * 10 + cu
cyc = d[:, 0, 0].double()
print("%d x %d x %d: span %.1f us; end time by XCD (mean / max, us) and lifetime cycles (mean)" % (m, n, k, float(end.max())))
for xc in sorted(set(xcc.tolist())):
    sel = xcc == xc
    print("  XCD %d: %3d workgroups (blockIdx %% 8 = %s)  end %7.1f / %7.1f   cycles %.0f   in-kernel clock %.3f GHz"
          % (xc, int(sel.sum()), sorted(set((torch.arange(512)[sel] % 8).tolist())), float(end[sel].mean()), float(end[sel].max()),
             float(cyc[sel].mean()), float((cyc[sel] / (r1[sel] - r0[sel])).mean()) * 0.1))
pairs = {}
for b in range(512):
    pairs.setdefault(int(cuid[b]), []).append(b)
sizes = [len(v) for v in pairs.values()]
print("CUs used: %d; workgroups per CU: min %d max %d" % (len(pairs), min(sizes), max(sizes)))
first, second, roles = [], [], []
for v in pairs.values():
    if len(v) == 2:
        a, b = sorted(v, key=lambda q: float(start[q]))
        first.append(float(end[a])); second.append(float(end[b]))
        roles.append((a >= 256, b >= 256))
if first:
    f, s2 = torch.tensor(first), torch.tensor(second)
    print("pairs: %d; end of the workgroup that started first %.1f us (mean), of the other %.1f us; |difference| mean %.1f us, "
          "later one is the second-started in %.0f %% of the CUs" % (len(first), float(f.mean()), float(s2.mean()),
                                                                   float((f - s2).abs().mean()), 100 * float((s2 > f).float().mean())))
    print("first-started has blockIdx < 256 and the other >= 256 in %d of %d CUs" % (sum(1 for r in roles if r == (False, True)), len(roles)))
    cu_end = torch.maximum(f, s2)
    print("CU finish time: mean %.1f, min %.1f, max %.1f us" % (float(cu_end.mean()), float(cu_end.min()), float(cu_end.max())))
