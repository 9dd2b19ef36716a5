#!/bin/bash
# round 6, call 5: the BatchNorm layers of the 16-bit path without their fp32 intermediate -- kernel tests, model tests, configs[2] A/B
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_gemm_h.py -m gpu -q -x > gpurun_out/c5_gemm_h.log 2>&1; rc=$?; tail -n 12 gpurun_out/c5_gemm_h.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 900 python -m pytest tests/test_gpu_float.py tests/test_gpu_model.py -m gpu -q -x -k "bf16 or fp16 or 16" > gpurun_out/c5_models.log 2>&1; rc=$?; tail -n 12 gpurun_out/c5_models.log
[ $rc -eq 0 ] || exit $rc
export BENCH_ARGS="--baseline-config 2 --steps 12 --warmup 3 --no-knn-check"
bash tools/ab_env.sh "c2_old:CCN_FUSE16=0" "c2_fuse512:CCN_FUSE16=1" "c2_fuse256:CCN_FUSE16_MAX=256" "c2_fuse1024:CCN_FUSE16_MAX=1024" "c2_old2:CCN_FUSE16=0"
