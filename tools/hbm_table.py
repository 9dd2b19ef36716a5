"""Per-kernel HBM GB/s (and MFMA pipe utilisation) table from a rocprofv3 kernel-stats CSV and a pmc_summary JSON.
    python tools/hbm_table.py profiles/r01f_kitti_kernel_stats.csv profiles/r01f_kitti_pmc.json ["bench.py flags"] > profiles/r01f_kitti_hbm_table.md"""
import csv
import json
import re
import sys

stats = {r["Name"]: r for r in csv.DictReader(open(sys.argv[1]))}
pmc = json.load(open(sys.argv[2]))
rows = []
for name, t in pmc.items():
    st = stats.get(name)
    if not st:
        continue
    avg_us = float(st["AverageNs"]) / 1e3
    rd, wr = t["fetch_bytes_per_launch_corrected"], t["write_bytes_per_launch"]
    short = re.sub(r"\(.*", "", name.replace("(anonymous namespace)::", "").replace("void ", ""))
    rows.append((float(st["Percentage"]), short[:58], int(st["Calls"]), avg_us, rd / 1e6, wr / 1e6,
                 (rd + wr) / (avg_us * 1e-6) / 1e9, t.get("mfma_pipe_utilisation", 0.0), t.get("effective_clock_ghz", 0.0)))
rows.sort(reverse=True)
flags = sys.argv[3] if len(sys.argv) > 3 else ""
print("# HBM traffic and MFMA utilisation per kernel, %s (rocprofv3)\n" % ("bench.py " + flags if flags else "full KITTI bench"))
print("Sources: `%s` (--kernel-trace --stats of `python bench.py%s`) and `%s` (separate `--pmc FETCH_SIZE`, `--pmc WRITE_SIZE` and"
      % (sys.argv[1], " " + flags if flags else "", sys.argv[2]))
print("`--pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE` passes of the same command; FETCH_SIZE doubled as")
print("MI355X_MICROARCH.md prescribes for gfx950).  GB/s = (read + written bytes per launch) / average launch duration; HBM3E")
print("spec peak 8000 GB/s.  MFMA util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8); clock = GRBM_GUI_ACTIVE")
print("/ 8 / duration in the (serialised) counter pass.  Durations come from the un-instrumented stats run, where kernels of")
print("two streams overlap (position-only work of the next step, everything else): a kernel's GB/s here is what it")
print("gets while sharing the chip, not its stand-alone rate (BN backward apply: 5.6 TB/s alone, 2.8 TB/s next to a GEMM).\n")
print("| % GPU time | kernel | launches | avg us | read MB | written MB | GB/s | % of 8 TB/s | MFMA util | clock GHz |")
print("|---|---|---|---|---|---|---|---|---|---|")
for pct, short, calls, avg, rd, wr, gbs, util, clk in rows[:32]:
    print("| %.1f | `%s` | %d | %.1f | %.1f | %.1f | %.0f | %.0f | %s | %.2f |"
          % (pct, short, calls, avg, rd, wr, gbs, 100 * gbs / 8000, ("%.2f" % util) if util > 0 else "-", clk))
