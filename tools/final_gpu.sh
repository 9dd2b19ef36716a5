#!/bin/bash
# Run ON the GPU box: the rocprofv3 evidence of a round for one workload -> gpurun_out/<tag>[_cN]_*
#   bash tools/final_gpu.sh <tag>            full KITTI bench (stats + three PMC passes)
#   bash tools/final_gpu.sh <tag> 2 4        ... of bench.py --baseline-config 2 and 4
# (the bench lines themselves: tools/final_round.sh; the HBM tables are made from the copies under profiles/ with tools/hbm_table.py)
tag=${1:-rXX}; shift
repo=${GRAFT_REPO_ROOT:-/root/repo}; out=$repo/gpurun_out
cd $repo
if [ $# -eq 0 ]; then
  bash tools/collect_profiles.sh $tag || exit 1
else
  for c in "$@"; do
    bash tools/collect_profiles.sh ${tag}_c$c --baseline-config $c || exit 1
    mv $out/${tag}_c${c}_kitti_kernel_stats.csv $out/${tag}_c${c}_kernel_stats.csv
    mv $out/${tag}_c${c}_kitti_pmc.json $out/${tag}_c${c}_pmc.json
  done
fi
echo "profiles done: $tag $*" >> $out/${tag}_progress.log
