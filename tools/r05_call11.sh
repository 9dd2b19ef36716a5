#!/bin/bash
# round 5: counted epilogue waits in the 4-wave persistent fp32 GEMM (N <= 64): GEMM / conv / model tests, one-box A/B of the step
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_float.py tests/test_gpu_gemm_f64.py tests/test_gpu_golden.py tests/test_gpu_contract.py "tests/test_gpu_model.py::test_model_forward_backward_matches_oracle" "tests/test_gpu_model.py::test_full_kitti_config_matches_oracle" "tests/test_gpu_model.py::test_deferred_activations_give_the_same_bits" -m gpu -q -x --timeout 600 > gpurun_out/pytest_c11.log 2>&1
rc=$?; tail -n 4 gpurun_out/pytest_c11.log | cut -c1-300; echo "pytest rc=$rc"
if [ $rc -ne 0 ]; then grep -n "^E  " gpurun_out/pytest_c11.log | head -20 | cut -c1-300; exit $rc; fi
PREV=$PWD/curvecloudnet_amd/libccn_hip_r05e.so
BENCH_ARGS="--steps 16 --no-second-line" tools/ab_env.sh "prev:CCN_LIB_PATH=$PREV" "new:CCN_NOTHING=1" "prev2:CCN_LIB_PATH=$PREV" "new2:CCN_NOTHING=1"
grep -E "persistent_kernel|conv_rows" gpurun_out/ab_prev_kernels.txt | sed 's/^/prev /'; grep -E "persistent_kernel|conv_rows" gpurun_out/ab_new_kernels.txt | sed 's/^/new  /'
grep -E "N=64 |N=32 " gpurun_out/ab_prev_shapes.txt | head -8 | sed 's/^/prev /'; grep -E "N=64 |N=32 " gpurun_out/ab_new_shapes.txt | head -8 | sed 's/^/new  /'
