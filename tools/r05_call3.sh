#!/bin/bash
# round 5: atomics-free backward of the compact first SGCNN layer: tests, then a one-box A/B of the whole step
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests/test_gpu_float.py tests/test_gpu_golden.py "tests/test_gpu_model.py::test_model_forward_backward_matches_oracle" "tests/test_gpu_model.py::test_full_kitti_config_matches_oracle" -m gpu -q -x --timeout 600 > gpurun_out/pytest_c3.log 2>&1
rc=$?; tail -n 6 gpurun_out/pytest_c3.log | cut -c1-300; echo "pytest rc=$rc"
if [ $rc -ne 0 ]; then grep -n "^E  " gpurun_out/pytest_c3.log | head -20 | cut -c1-300; exit $rc; fi
BENCH_ARGS="--steps 16 --no-second-line" tools/ab_env.sh "atomics:CCN_CG_BWD_GATHER=0 CCN_PN_BWD_GATHER=0" "cg:CCN_PN_BWD_GATHER=0" "cgpn:CCN_PN_BWD_GATHER=1" "atomics2:CCN_CG_BWD_GATHER=0 CCN_PN_BWD_GATHER=0" "cgpn2:CCN_PN_BWD_GATHER=1"
grep -E "pn_edge" gpurun_out/ab_cg_kernels.txt gpurun_out/ab_cgpn_kernels.txt
