#!/bin/bash
# Run ON the GPU box: SQ counters of the paired GEMM kernel alone -> gpurun_out/<tag>_pair_pmc.txt
tag=${1:-rXX}; repo=${GRAFT_REPO_ROOT:-/root/repo}; out=$repo/gpurun_out
cd /tmp && export TMPDIR=/tmp
: > $out/${tag}_pair_pmc.txt
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_INSTS_LDS SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_LEVEL_LDS GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d /tmp/pp_$i -o c -- python3 $repo/tools/pair_one.py 1342781 256 256 > $out/${tag}_pair_pmc_$i.log 2>&1 || { echo "pass $i failed" >> $out/${tag}_pair_pmc.txt; continue; }
  python3 - "$(find /tmp/pp_$i -name '*counter_collection.csv' | head -1)" >> $out/${tag}_pair_pmc.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(float); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if "gemm_glds_pair" in r["Kernel_Name"]:
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in acc:
    print("%-32s %16.0f per launch (%d launches)" % (k, acc[k] / n[k], n[k]))
PY
done
cat $out/${tag}_pair_pmc.txt
