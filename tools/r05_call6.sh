#!/bin/bash
# round 5: inverse row lists built by own kernels (no torch.sort / bincount on the geometry stream): tests, one-box A/B of the step
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests/test_gpu_index.py tests/test_gpu_float.py tests/test_gpu_golden.py "tests/test_gpu_model.py::test_model_forward_backward_matches_oracle" -m gpu -q -x --timeout 600 > gpurun_out/pytest_c6.log 2>&1
rc=$?; tail -n 6 gpurun_out/pytest_c6.log | cut -c1-300; echo "pytest rc=$rc"
if [ $rc -ne 0 ]; then grep -n "^E  " gpurun_out/pytest_c6.log | head -20 | cut -c1-300; exit $rc; fi
BENCH_ARGS="--steps 16 --no-second-line" tools/ab_env.sh "torchsort:CCN_INV_TORCH=1" "own:CCN_NOTHING=1" "torchsort2:CCN_INV_TORCH=1" "own2:CCN_NOTHING=1"
grep -E "inverse_lists|group_owner" gpurun_out/ab_own_kernels.txt
