"""Fold rocprofv3 --pmc outputs (one *_counter_collection.csv per pass) into per-kernel JSON.

    python tools/pmc_summary.py OUT.json fetch=<csv> write=<csv> [mfma=<csv>]

fetch / write: FETCH_SIZE and WRITE_SIZE passes (collected separately: they do not fit one pass).  rocprofv3 reports
both in KiB (request count x 64 B / 1024); on gfx950 FETCH_SIZE reads half of the bytes of wide streaming reads
(MI355X_MICROARCH.md, HBM section), so it is doubled here.
mfma: a pass with SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES and GRBM_GUI_ACTIVE.
"""
import csv
import json
import sys
from collections import defaultdict


def fold(path):
    per = defaultdict(lambda: defaultdict(float))
    launches = defaultdict(set)
    dur = defaultdict(float)
    seen = set()
    with open(path) as f:
        for r in csv.DictReader(f):
            k = r["Kernel_Name"]
            per[k][r["Counter_Name"]] += float(r["Counter_Value"])
            launches[k].add(r["Dispatch_Id"])
            key = (k, r["Dispatch_Id"])
            if key not in seen and r.get("End_Timestamp") and r.get("Start_Timestamp"):
                seen.add(key)
                dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    return per, {k: len(v) for k, v in launches.items()}, dur


def main():
    out = sys.argv[1]
    passes = dict(a.split("=", 1) for a in sys.argv[2:])
    table = {}
    if "fetch" in passes:
        per, n, _ = fold(passes["fetch"])
        for k, c in per.items():
            t = table.setdefault(k, {})
            t["launches"] = n[k]
            t["fetch_bytes_per_launch_raw"] = c.get("FETCH_SIZE", 0.0) * 1024.0 / n[k]
            t["fetch_bytes_per_launch_corrected"] = 2.0 * t["fetch_bytes_per_launch_raw"]
    if "write" in passes:
        per, n, _ = fold(passes["write"])
        for k, c in per.items():
            t = table.setdefault(k, {})
            t.setdefault("launches", n[k])
            t["write_bytes_per_launch"] = c.get("WRITE_SIZE", 0.0) * 1024.0 / n[k]
    if "mfma" in passes:
        per, n, dur = fold(passes["mfma"])
        for k, c in per.items():
            t = table.setdefault(k, {})
            t.setdefault("launches", n[k])
            busy, gui = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), c.get("GRBM_GUI_ACTIVE", 0.0)
            t["mfma_busy_cycles_per_launch"] = busy / n[k]
            t["gui_active_per_launch"] = gui / n[k]
            t["sq_busy_cycles_per_launch"] = c.get("SQ_BUSY_CYCLES", 0.0) / n[k]
            if dur[k] > 0 and gui > 0:
                t["effective_clock_ghz"] = gui / 8.0 / dur[k]          # GUI_ACTIVE is summed over the 8 XCDs
                # SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs; the chip ran gui/8 cycles
                t["mfma_pipe_utilisation"] = busy / (1024.0 * gui / 8.0)
    for t in table.values():
        t.setdefault("fetch_bytes_per_launch_corrected", 0.0)
        t.setdefault("write_bytes_per_launch", 0.0)
    json.dump(table, open(out, "w"), indent=1, sort_keys=True)
    print("wrote %s: %d kernels" % (out, len(table)))


if __name__ == "__main__":
    main()
