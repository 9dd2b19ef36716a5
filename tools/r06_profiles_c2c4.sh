#!/bin/bash
# rocprofv3 stats + PMC passes of BASELINE configs[2] and configs[4] (tools/collect_profiles.sh writes <tag>_kitti_* names: renamed here)
for c in 2 4; do
  bash tools/collect_profiles.sh r06c$c --baseline-config $c || exit 1
  mv gpurun_out/r06c${c}_kitti_kernel_stats.csv gpurun_out/r06_c${c}_kernel_stats.csv
  mv gpurun_out/r06c${c}_kitti_pmc.json gpurun_out/r06_c${c}_pmc.json
done
