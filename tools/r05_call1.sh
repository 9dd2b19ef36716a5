#!/bin/bash
# round 5, first GPU call: the new kernels' tests, the split microbench, a one-box A/B of the split, the default bench line
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_gemm_split.py tests/test_gpu_frnn_compat.py tests/test_gpu_gemm_f64.py tests/test_gpu_contract.py -m gpu -q -x --timeout 400 -s > gpurun_out/pytest_c1.log 2>&1
rc=$?; tail -n 15 gpurun_out/pytest_c1.log; echo "pytest rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 200 python tools/bench_split.py > gpurun_out/bench_split.txt 2>&1 && cat gpurun_out/bench_split.txt &&
BENCH_ARGS="--steps 16 --no-second-line" tools/ab_env.sh "nosplit:CCN_NT_SPLIT=0" "split:CCN_NT_SPLIT=1" &&
timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_c1.json 2> gpurun_out/bench_c1.err
brc=$?; tail -c 600 gpurun_out/bench_c1.err; head -c 2500 gpurun_out/bench_c1.json; echo "bench rc=$brc"
cp gpurun_out/bench_kernels.txt gpurun_out/bench_kernels_c1.txt; cp gpurun_out/bench_gemm_shapes.txt gpurun_out/bench_gemm_shapes_c1.txt
exit $brc
