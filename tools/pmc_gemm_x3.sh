#!/bin/bash
# PMC passes over one GEMM shape for the split-bf16 kernel: tools/pmc_gemm_x3.sh M K N
cd /tmp && export TMPDIR=/tmp
repo=${GRAFT_REPO_ROOT:-/root/repo}
export PYTHONPATH=$repo
out=$repo/gpurun_out
M=$1; K=$2; N=$3
i=0
for counters in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_MFMA"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $counters --output-format csv -d /tmp/px3_$i -o c -- python3 $repo/tools/gemm_one_shape.py $M $K $N 5 gemm_nt_x3 > $out/pmc_x3_$i.log 2>&1 || { echo "pass $i failed"; tail -5 $out/pmc_x3_$i.log; exit 1; }
  f=$(find /tmp/px3_$i -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "x3" not in k: continue
    k = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-60:]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
for k, d in acc.items():
    print(k, {c: "%.4g" % v for c, v in d.items()})
PY
done
