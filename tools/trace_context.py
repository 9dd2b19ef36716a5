"""Kernels around the largest idle gaps of a rocprofv3 --kernel-trace CSV (all queues): python tools/trace_context.py <csv> [n_gaps] [context]"""
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
rows.sort()
ngaps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
ctx = int(sys.argv[3]) if len(sys.argv) > 3 else 6
t0 = rows[0][0]
rows = [r for r in rows if r[0] >= t0 + 0.6 * (rows[-1][0] - t0)]
gaps, cur_e = [], rows[0][1]
for idx in range(1, len(rows)):
    s, e = rows[idx][0], rows[idx][1]
    if s > cur_e:
        gaps.append((s - cur_e, idx))
    cur_e = max(cur_e, e)
gaps.sort(reverse=True)


def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]


for gap, idx in gaps[:ngaps]:
    print("---- gap %.2f ms" % (gap / 1e6))
    for k in range(max(0, idx - ctx), min(len(rows), idx + ctx)):
        s, e, n, q = rows[k]
        print("%s q%s  +%9.3f ms  dur %8.3f ms  %s" % (">>" if k == idx else "  ", q, (s - rows[idx][0]) / 1e6, (e - s) / 1e6, short(n)))
