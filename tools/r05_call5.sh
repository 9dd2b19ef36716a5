#!/bin/bash
# round 5: batched (several points per wave) form of the compact SGCNN first-layer kernels: tests, then a one-box A/B of the step
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests/test_gpu_float.py tests/test_gpu_golden.py tests/test_gpu_gemm_h.py "tests/test_gpu_model.py::test_model_forward_backward_matches_oracle" "tests/test_gpu_model.py::test_full_kitti_config_matches_oracle" -m gpu -q -x --timeout 600 > gpurun_out/pytest_c5.log 2>&1
rc=$?; tail -n 6 gpurun_out/pytest_c5.log | cut -c1-300; echo "pytest rc=$rc"
if [ $rc -ne 0 ]; then grep -n "^E  " gpurun_out/pytest_c5.log | head -20 | cut -c1-300; exit $rc; fi
PREV=$PWD/curvecloudnet_amd/${PREV_LIB:-libccn_hip_r05b.so}
BENCH_ARGS="--steps 16 --no-second-line" tools/ab_env.sh "prev:CCN_LIB_PATH=$PREV" "new:CCN_NOTHING=1" "prev2:CCN_LIB_PATH=$PREV" "new2:CCN_NOTHING=1"
for k in ${KERNELS:-cg_edge cg_max}; do grep -E "$k" gpurun_out/ab_prev_kernels.txt | sed 's/^/prev /'; grep -E "$k" gpurun_out/ab_new_kernels.txt | sed 's/^/new  /'; done
