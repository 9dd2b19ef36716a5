#!/bin/bash
# Run ON the GPU box: two rocprofv3 --stats runs of bench.py (4 and 12 timed steps) -> gpurun_out/<tag>_per_step.txt
tag=${1:-rXX}; shift
repo=${GRAFT_REPO_ROOT:-/root/repo}; out=$repo/gpurun_out
cd /tmp && export TMPDIR=/tmp
for s in 4 12; do
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps_$s -o s -- python3 $repo/bench.py --no-cpu-baseline --no-kernel-timing --warmup 2 --steps $s $* > $out/${tag}_ps_${s}.log 2>&1 || exit 1
  cp "$(find /tmp/ps_$s -name '*kernel_stats.csv' | head -1)" $out/${tag}_stats_${s}steps.csv
done
python3 $repo/tools/per_step_kernels.py $out/${tag}_stats_4steps.csv 4 $out/${tag}_stats_12steps.csv 12 > $out/${tag}_per_step.txt
head -n 3 $out/${tag}_per_step.txt
