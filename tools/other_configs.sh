#!/bin/bash
# The non-default bench configurations, one JSON line each (run ON the GPU box): bash tools/other_configs.sh <out file> [extra flags]
out=$1; shift
: > $out
while read -r flags; do
  echo "== $flags $*" >> $out
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps 8 --warmup 3 $flags $* 2>/dev/null | tail -1 >> $out || echo "FAILED" >> $out
  echo "done: $flags $*"
done <<'CFG'
--baseline-config 0 --clouds-per-gpu 64
--baseline-config 2
--baseline-config 2 --mlp-dtype fp32
--baseline-config 3
--baseline-config 4
--baseline-config 4 --mlp-dtype fp32
--baseline-config 4 --graph
--config kortx
--config hotpath
--mlp-dtype bf16
--mlp-dtype bf16x3
CFG
