#!/bin/bash
# rocprofv3 stats + PMC passes of BASELINE configs[4]; a heartbeat keeps gpurun's silence watchdog quiet (the PMC passes of this
# configuration serialise a 12 k-round FPS chain per step and print nothing for minutes)
( while true; do date >> gpurun_out/r06c4_heartbeat.log; sleep 60; done ) &
hb=$!
bash tools/collect_profiles.sh r06c4 --baseline-config 4; rc=$?
kill $hb
[ $rc -eq 0 ] || exit $rc
mv gpurun_out/r06c4_kitti_kernel_stats.csv gpurun_out/r06_c4_kernel_stats.csv
mv gpurun_out/r06c4_kitti_pmc.json gpurun_out/r06_c4_pmc.json
