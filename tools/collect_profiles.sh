#!/bin/bash
# Run ON the GPU box (under gpurun): rocprofv3 kernel statistics + separate PMC passes of the default bench command.
# Usage: bash tools/collect_profiles.sh <tag> [extra bench.py flags]   -> gpurun_out/<tag>_*  (copy what should be judged
# into profiles/), e.g. "r01n_x3 --mlp-dtype bf16x3"
set -o pipefail
tag=${1:-rXX}
shift
repo=${GRAFT_REPO_ROOT:-/root/repo}
out=$repo/gpurun_out
cd /tmp && export TMPDIR=/tmp
bench="python3 $repo/bench.py --no-cpu-baseline --no-kernel-timing --no-second-line $*"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats -o s -- $bench > $out/${tag}_stats_run.log 2>&1 || exit 1
cp "$(find /tmp/p_stats -name '*kernel_stats.csv' | head -1)" $out/${tag}_kitti_kernel_stats.csv
for pass in fetch:FETCH_SIZE write:WRITE_SIZE "mfma:SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  name=${pass%%:*}; counters=${pass#*:}
  timeout -k 10 500 rocprofv3 --pmc $counters --output-format csv -d /tmp/p_$name -o c -- $bench --steps 3 --warmup 2 > $out/${tag}_pmc_${name}_run.log 2>&1 || exit 1
  echo "pass $name done" >> $out/${tag}_progress.log
done
python3 $repo/tools/pmc_summary.py $out/${tag}_kitti_pmc.json \
  fetch="$(find /tmp/p_fetch -name '*counter_collection.csv' | head -1)" \
  write="$(find /tmp/p_write -name '*counter_collection.csv' | head -1)" \
  mfma="$(find /tmp/p_mfma -name '*counter_collection.csv' | head -1)"
