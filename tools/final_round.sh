#!/bin/bash
# Run ON the GPU box: the artefacts of a round in one call -> gpurun_out/<tag>_*
#   default bench line (with cpu baseline + kernel tables), the same with the weight-gradient stream on, other configs
tag=${1:-rXX}
repo=${GRAFT_REPO_ROOT:-/root/repo}; out=$repo/gpurun_out
cd $repo
timeout -k 10 600 python bench.py > $out/${tag}_kitti_bench.json 2> $out/${tag}_bench.err || exit 1
cp $out/bench_kernels.txt $out/${tag}_kitti_bench_kernels.txt; cp $out/bench_gemm_shapes.txt $out/${tag}_kitti_gemm_shapes.txt
cp $out/bench_kernels_second_line.txt $out/${tag}_kitti_bf16x3_bench_kernels.txt 2>/dev/null   # the second line's own kernel table
echo "bench done" >> $out/${tag}_progress.log
timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/${tag}_kitti_bench_driver_flags.json 2>> $out/${tag}_bench.err || exit 1
echo "driver-flags bench done" >> $out/${tag}_progress.log
CCN_WGRAD_STREAM=1 timeout -k 10 400 python bench.py --no-cpu-baseline --no-second-line > $out/${tag}_kitti_bench_wgrad_stream.json 2>> $out/${tag}_bench.err || exit 1
echo "ws done" >> $out/${tag}_progress.log
bash tools/other_configs.sh $out/${tag}_other_configs_bench.txt
