#!/bin/bash
# round 5: split tails with agent-scope partials: tests, microbench, one-box A/B against the round-4 library
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_gemm_split.py -m gpu -q -x --timeout 200 > gpurun_out/pytest_c2.log 2>&1
rc=$?; tail -n 5 gpurun_out/pytest_c2.log; echo "pytest rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 200 python tools/bench_split.py > gpurun_out/bench_split.txt 2>&1 && cat gpurun_out/bench_split.txt &&
BENCH_ARGS="--steps 16 --no-second-line" tools/ab_env.sh "r04lib:CCN_NT_SPLIT=0 CCN_LIB_PATH=$PWD/curvecloudnet_amd/libccn_hip_r04.so" "nosplit:CCN_NT_SPLIT=0" "split:CCN_NT_SPLIT=1" "r04lib2:CCN_NT_SPLIT=0 CCN_LIB_PATH=$PWD/curvecloudnet_amd/libccn_hip_r04.so"
