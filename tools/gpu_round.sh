#!/bin/bash
# One gpurun call: GPU tests (log under gpurun_out/), then -- unless the tests were KILLED at a limit -- a short bench.
# usage: tools/gpu_round.sh <tag> [pytest args...]
tag=$1; shift
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q --timeout 900 "$@" > gpurun_out/pytest_$tag.log 2>&1
rc=$?
tail -n 40 gpurun_out/pytest_$tag.log
echo "pytest rc=$rc"
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
timeout -k 10 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err
brc=$?
tail -c 1500 gpurun_out/bench_$tag.err
cat gpurun_out/bench_$tag.json | head -c 3000
cp gpurun_out/bench_kernels.txt gpurun_out/bench_kernels_$tag.txt 2>/dev/null
cp gpurun_out/bench_gemm_shapes.txt gpurun_out/bench_gemm_shapes_$tag.txt 2>/dev/null
echo "bench rc=$brc"
exit $rc
