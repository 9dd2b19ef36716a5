#!/bin/bash
# round 5: XCD-aware work order in the row-gathering kernels: tests of the kernels concerned, then a one-box A/B of the step
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests/test_gpu_float.py tests/test_gpu_golden.py tests/test_gpu_contract.py "tests/test_gpu_model.py::test_model_forward_backward_matches_oracle" -m gpu -q -x --timeout 600 > gpurun_out/pytest_c4.log 2>&1
rc=$?; tail -n 6 gpurun_out/pytest_c4.log | cut -c1-300; echo "pytest rc=$rc"
if [ $rc -ne 0 ]; then grep -n "^E  " gpurun_out/pytest_c4.log | head -20 | cut -c1-300; exit $rc; fi
PREV=$PWD/curvecloudnet_amd/libccn_hip_r05a.so
BENCH_ARGS="--steps 16 --no-second-line" tools/ab_env.sh "idorder:CCN_LIB_PATH=$PREV" "xcd:CCN_NOTHING=1" "idorder2:CCN_LIB_PATH=$PREV" "xcd2:CCN_NOTHING=1"
for k in cg_edge pn_edge cg_max seg_softmax interp gather_rows edge_feat; do grep -E "$k" gpurun_out/ab_idorder_kernels.txt | sed 's/^/id  /'; grep -E "$k" gpurun_out/ab_xcd_kernels.txt | sed 's/^/xcd /'; done
