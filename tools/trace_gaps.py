"""Summarise a rocprofv3 --kernel-trace CSV: GPU busy time (union of kernel intervals), idle gaps and the
kernels that precede the largest gaps.  Usage: python tools/trace_gaps.py <kernel_trace.csv> [last_fraction]"""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
rows.sort()
t0, t1 = rows[0][0], max(r[1] for r in rows)
cut = t1 - int((t1 - t0) * frac)                      # the steady-state tail of the run
rows = [r for r in rows if r[0] >= cut]
span = max(r[1] for r in rows) - rows[0][0]
busy, cur_s, cur_e = 0, rows[0][0], rows[0][1]
gaps = []
prev_name = rows[0][2]
for s, e, name, q in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, prev_name, name))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
    if e >= cur_e:
        prev_name = name
busy += cur_e - cur_s
per_queue = defaultdict(int)
for s, e, name, q in rows:
    per_queue[q] += e - s
print("window %.1f ms, busy %.1f ms (%.1f%%), idle %.1f ms in %d gaps" % (span / 1e6, busy / 1e6, 100.0 * busy / span,
                                                                          (span - busy) / 1e6, len(gaps)))
print("kernel time per queue (ms):", {q: round(v / 1e6, 1) for q, v in per_queue.items()})
hist = defaultdict(lambda: [0, 0])
for g, a, b in gaps:
    k = (a[:50], b[:50])
    hist[k][0] += g
    hist[k][1] += 1
print("largest idle contributors (after kernel -> before kernel):")
for (a, b), (tot, cnt) in sorted(hist.items(), key=lambda kv: -kv[1][0])[:25]:
    print("  %8.2f ms %5d x  %s -> %s" % (tot / 1e6, cnt, a, b))
buckets = [0, 0, 0, 0]
for g, _, _ in gaps:
    buckets[0 if g < 5e3 else 1 if g < 5e4 else 2 if g < 5e5 else 3] += g
print("idle by gap size: <5us %.1f ms, 5-50us %.1f ms, 50-500us %.1f ms, >500us %.1f ms" % tuple(b / 1e6 for b in buckets))

# ---- per-queue view: how long is the busiest (feature) queue idle, and what runs on the other queues meanwhile
by_q = defaultdict(list)
for s_, e_, name, q in rows:
    by_q[q].append((s_, e_, name))
main_q = max(by_q, key=lambda q: sum(e - s for s, e, _ in by_q[q]))
iv = sorted(by_q[main_q])
gaps_main = []
cur_e = iv[0][1]
for s_, e_, name in iv[1:]:
    if s_ > cur_e:
        gaps_main.append((cur_e, s_, name))
    cur_e = max(cur_e, e_)
tot_gap = sum(b - a for a, b, _ in gaps_main)
# who sits on either side of the feature queue's gaps
pair = defaultdict(lambda: [0, 0])
cur_e, cur_name = iv[0][1], iv[0][2]
for s_, e_, name in iv[1:]:
    if s_ > cur_e:
        k = (cur_name[:48], name[:48])
        pair[k][0] += s_ - cur_e
        pair[k][1] += 1
    if e_ >= cur_e:
        cur_e, cur_name = e_, name
print("feature-queue gaps by (kernel before -> kernel after):")
for (a_, b_), (tot_, cnt_) in sorted(pair.items(), key=lambda kv: -kv[1][0])[:30]:
    print("  %8.2f ms %5d x  %s -> %s" % (tot_ / 1e6, cnt_, a_, b_))
sizes = [0, 0, 0, 0]
for a_, b_, _ in gaps_main:
    g_ = b_ - a_
    sizes[0 if g_ < 5e3 else 1 if g_ < 5e4 else 2 if g_ < 5e5 else 3] += g_
print("feature-queue idle by gap size: <5us %.1f ms, 5-50us %.1f ms, 50-500us %.1f ms, >500us %.1f ms" % tuple(b_ / 1e6 for b_ in sizes))
print("feature queue %s: idle %.1f ms of %.1f ms window (%d gaps)" % (main_q, tot_gap / 1e6, span / 1e6, len(gaps_main)))
others = sorted((s_, e_, name) for q in by_q if q != main_q for s_, e_, name in by_q[q])
blame = defaultdict(float)
for a, b, nxt in gaps_main:
    if b - a < 20e3:
        continue
    for s_, e_, name in others:
        if e_ <= a:
            continue
        if s_ >= b:
            break
        blame[name[:60]] += min(e_, b) - max(s_, a)
print("kernels on the other queues running during feature-queue gaps > 20 us:")
for name, t_ in sorted(blame.items(), key=lambda kv: -kv[1])[:15]:
    print("  %8.2f ms  %s" % (t_ / 1e6, name))

# ---- timeline of the last large feature-queue gap: everything that runs on any queue from 2 ms before it to its end
big = [g for g in gaps_main if g[1] - g[0] > 2e6]
if big:
    a, b, nxt = big[-1]
    print("timeline around the last feature-queue gap > 2 ms (%.2f ms, before %s); times in ms relative to the gap start:"
          % ((b - a) / 1e6, nxt[:40]))
    for s_, e_, name, q in rows:
        if e_ >= a - 2e6 and s_ <= b + 2e5:
            print("  q%-2s %9.3f .. %9.3f  %s" % (q, (s_ - a) / 1e6, (e_ - a) / 1e6, name[:90]))

# ---- coarse timeline of the last full step (between the last two optimizer launches): per queue, runs of kernels whose
# gaps are below 150 us, with the first and last kernel of each run
adams = [s_ for s_, e_, name, q in rows if "adam_kernel" in name]
marks = [adams[0]] if adams else []
for t_ in adams[1:]:
    if t_ - marks[-1] > 20e6:
        marks.append(t_)
if len(marks) >= 2:
    lo_, hi_ = marks[-2], marks[-1]
    print("coarse timeline of the last step (%.1f ms between optimizer launches), ms from the first:" % ((hi_ - lo_) / 1e6))
    for q in sorted(by_q):
        runs = []
        for s_, e_, name in sorted(by_q[q]):
            if e_ < lo_ or s_ > hi_:
                continue
            if runs and s_ - runs[-1][1] < 150e3:
                runs[-1][1] = max(runs[-1][1], e_)
                runs[-1][3] = name
                runs[-1][4] += 1
            else:
                runs.append([s_, e_, name, name, 1])
        for s_, e_, first, last, cnt in runs:
            print("  q%-2s %8.2f .. %8.2f  %5d kernels  %s ... %s" % (q, (s_ - lo_) / 1e6, (e_ - lo_) / 1e6, cnt, first[:40], last[:40]))
