#!/bin/bash
# round 6, call 4: per-launch records of two instrumented steps (name, family, ms, floor ms, GFLOP, MB, integer arguments)
CCN_BENCH_DUMP_RECORDS=$PWD/gpurun_out/r06_records.tsv timeout -k 10 300 python bench.py --steps 8 --warmup 3 --no-second-line --no-knn-check --no-cpu-baseline > gpurun_out/r06_rec.json 2> gpurun_out/r06_rec.err
wc -l gpurun_out/r06_records.tsv
