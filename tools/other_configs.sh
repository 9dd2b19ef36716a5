#!/bin/bash
# The non-default bench configurations, one JSON line each (run ON the GPU box): bash tools/other_configs.sh <out file> [extra flags]
out=$1; shift
: > $out
while read -r flags; do
  echo "== $flags $*" >> $out
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-kernel-timing --steps 8 --warmup 3 $flags $* 2>/dev/null | tail -1 >> $out || echo "FAILED" >> $out
  echo "done: $flags $*"
done <<'CFG'
--config nuscenes --curves 1430 --clouds-per-gpu 16
--config kitti --curves 4900 --clouds-per-gpu 4
--config a2d2 --mixed-lengths --clouds-per-gpu 8
--config shapenet-seg --curves 84 --clouds-per-gpu 64
--config kortx
--config hotpath
CFG
