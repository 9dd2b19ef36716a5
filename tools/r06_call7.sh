#!/bin/bash
# round 6, call 7: softmax-aggregation backward with the group kept in registers -- tests, then the step
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_float.py tests/test_gpu_gemm_h.py tests/test_gpu_golden.py tests/test_gpu_contract.py -m gpu -q -x -k "softmax or attend or sa_ or pointnet or PointNet or step_modules or edge_reduce" > gpurun_out/c7_tests.log 2>&1; rc=$?; tail -n 6 gpurun_out/c7_tests.log
[ $rc -eq 0 ] || exit $rc
export BENCH_ARGS="--steps 16 --warmup 3 --no-second-line --no-knn-check"
bash tools/ab_env.sh "new:CCN_X=0" "new2:CCN_X=0"
grep -h "seg_softmax" gpurun_out/ab_new_kernels.txt
