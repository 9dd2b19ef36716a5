"""Per-step view of a rocprofv3 --kernel-trace CSV of bench.py: steps are delimited by the first adam_kernel launch of each
optimiser step; for the middle steps: duration, busy time per queue, idle time of the main (feature) queue and what
precedes its largest gaps.    python tools/trace_steps.py <kernel_trace.csv>"""
import csv
import sys
from collections import defaultdict


def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:48]


rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "?")))
rows.sort()
adam = [r for r in rows if r[2].startswith("adam_kernel")]
marks = [adam[0][0]]
for a, b in zip(adam, adam[1:]):
    if b[0] - a[1] > 20e6:          # > 20 ms apart: a new optimiser step
        marks.append(b[0])
main_q = adam[0][3]
print("%d optimiser steps, main queue %s" % (len(marks), main_q))
for si in range(len(marks) // 2, min(len(marks) - 1, len(marks) // 2 + 3)):
    t0, t1 = marks[si], marks[si + 1]
    ks = [r for r in rows if t0 <= r[0] < t1]
    busy = defaultdict(int)
    for s, e, n, q in ks:
        busy[q] += e - s
    mq = [r for r in ks if r[3] == main_q]
    gaps, cur = [], mq[0][1]
    prev = mq[0][2]
    for s, e, n, q in mq[1:]:
        if s > cur:
            gaps.append((s - cur, prev, n))
        if e > cur:
            cur, prev = e, n
    idle = sum(g[0] for g in gaps)
    print("step %d: %.1f ms; busy per queue (ms): %s; main queue idle %.1f ms in %d gaps (<10us: %.1f, 10-100us: %.1f, >100us: %.1f)" % (
        si, (t1 - t0) / 1e6, {q: round(v / 1e6, 1) for q, v in sorted(busy.items())}, idle / 1e6, len(gaps),
        sum(g[0] for g in gaps if g[0] < 1e4) / 1e6, sum(g[0] for g in gaps if 1e4 <= g[0] < 1e5) / 1e6,
        sum(g[0] for g in gaps if g[0] >= 1e5) / 1e6))
    agg = defaultdict(lambda: [0, 0])
    for g, a, b in gaps:
        agg[(a, b)][0] += g
        agg[(a, b)][1] += 1
    for (a, b), (g, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:12]:
        print("    %7.2f ms %4d x  %s -> %s" % (g / 1e6, c, a, b))
