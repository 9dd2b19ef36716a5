"""Per-step kernel time from two rocprofv3 --stats runs of bench.py that differ only in --steps: everything that does not
repeat per step (model construction, warm-up, the kernel-table passes) cancels in the difference.
usage: python tools/per_step_kernels.py <stats_a.csv> <steps_a> <stats_b.csv> <steps_b> [substring filter]"""
import csv
import sys


def load(path):
    out = {}
    for r in csv.DictReader(open(path)):
        out[r["Name"]] = (int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6)
    return out


a, sa, b, sb = load(sys.argv[1]), int(sys.argv[2]), load(sys.argv[3]), int(sys.argv[4])
flt = sys.argv[5] if len(sys.argv) > 5 else ""
rows = []
for name in set(a) | set(b):
    ca, ta = a.get(name, (0, 0.0))
    cb, tb = b.get(name, (0, 0.0))
    rows.append(((tb - ta) / (sb - sa), (cb - ca) / (sb - sa), name))
total = sum(r[0] for r in rows)
lib = sum(r[0] for r in rows if any(t in r[2] for t in ("at::", "rocprim", "rocclr", "hipcub")))
print("per step: %.2f ms of kernel time, %.2f ms of it in library kernels (torch / rocprim / runtime copies)" % (total, lib))
for ms, calls, name in sorted(rows, reverse=True):
    if flt in name and (ms > 0.02 or calls > 0.5):
        print("%8.3f ms %7.1f x  %s" % (ms, calls, name[:170]))
