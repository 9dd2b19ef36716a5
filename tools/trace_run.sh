#!/bin/bash
# Run ON the GPU box: rocprofv3 kernel trace of a short bench run, idle-gap summary -> gpurun_out/<tag>_gaps.txt
tag=${1:-rXX}; shift
repo=${GRAFT_REPO_ROOT:-/root/repo}; out=$repo/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o t -- python3 $repo/bench.py --no-cpu-baseline --no-kernel-timing --warmup 2 --steps 6 $* > $out/${tag}_kt.log 2>&1 || exit 1
python3 $repo/tools/trace_gaps.py "$(find /tmp/kt -name '*kernel_trace.csv' | head -1)" 0.3 > $out/${tag}_gaps.txt
tail -n 52 $out/${tag}_gaps.txt
