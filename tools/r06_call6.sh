#!/bin/bash
# round 6, call 6: the stream-of-clouds graph test on its own (it replays a deliberately overflowing batch), then the 16-bit kernel tests and configs[2]
mkdir -p gpurun_out
timeout -k 10 240 python -m pytest tests/test_gpu_graph.py -m gpu -q -x -k "stream_of_different" > gpurun_out/c6_graph.log 2>&1; rc=$?; tail -n 15 gpurun_out/c6_graph.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python -m pytest tests/test_gpu_gemm_h.py tests/test_gpu_graph.py -m gpu -q -x > gpurun_out/c6_h.log 2>&1; rc=$?; tail -n 5 gpurun_out/c6_h.log
[ $rc -eq 0 ] || exit $rc
export BENCH_ARGS="--baseline-config 2 --steps 12 --warmup 3 --no-knn-check"
bash tools/ab_env.sh "c2_old:CCN_FUSE16=0" "c2_fuse:CCN_FUSE16=1" "c2_old2:CCN_FUSE16=0" "c2_fuse2:CCN_FUSE16=1"
