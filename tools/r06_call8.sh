#!/bin/bash
# round 6, call 8: ccn_cg_edge_apply with software-pipelined index loads over several points per wave -- tests, then A/B on the step
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_float.py tests/test_gpu_golden.py tests/test_gpu_contract.py tests/test_gpu_model.py -m gpu -q -x -k "sgcnn or compact or cg_ or step_modules or model_forward_backward or full_kitti_config" > gpurun_out/c8_tests.log 2>&1; rc=$?; tail -n 6 gpurun_out/c8_tests.log
[ $rc -eq 0 ] || exit $rc
export BENCH_ARGS="--steps 12 --warmup 3 --no-second-line --no-knn-check"
bash tools/ab_env.sh "pts1:CCN_CG_APPLY_PTS=1" "auto:CCN_X=0" "pts4:CCN_CG_APPLY_PTS=4" "pts8:CCN_CG_APPLY_PTS=8" "pts64:CCN_CG_APPLY_PTS=64" "pts1b:CCN_CG_APPLY_PTS=1"
for n in pts1 auto pts4 pts8 pts64 pts1b; do echo "$n $(grep -h ' cg_edge_apply' gpurun_out/ab_${n}_kernels.txt)"; done
