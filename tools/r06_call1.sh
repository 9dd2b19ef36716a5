#!/bin/bash
# round 6, call 1: the full-size fp64 adjudication with the tail split off / on, then a baseline bench line
mkdir -p gpurun_out; rm -f gpurun_out/parity_split_ab.txt
for s in 0 1; do
  echo "== CCN_NT_SPLIT=$s" >> gpurun_out/parity_split_ab.txt
  CCN_NT_SPLIT=$s CCN_PARITY_LOG=$PWD/gpurun_out/parity_split_ab.txt timeout -k 10 400 python -m pytest tests/test_gpu_model.py -m gpu -q -x \
     -k "test_full_width_kitti_backward_on_the_full_size_cloud or test_full_kitti_config_matches_oracle" > gpurun_out/parity_split_$s.log 2>&1 || { tail -n 30 gpurun_out/parity_split_$s.log; }
  tail -n 3 gpurun_out/parity_split_$s.log
done
cat gpurun_out/parity_split_ab.txt
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r06_base.json 2> gpurun_out/r06_base.err; tail -c 1500 gpurun_out/r06_base.json
