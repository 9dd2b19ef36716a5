#!/bin/bash
# round 5: exact FPS over clouds of more than 16 k points by a cluster of workgroups: bit-identity tests, then configs[4] A/B
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_index.py -m gpu -q -x --timeout 200 -k "fps" > gpurun_out/pytest_c8.log 2>&1
rc=$?; tail -n 4 gpurun_out/pytest_c8.log | cut -c1-300; echo "pytest rc=$rc"
if [ $rc -ne 0 ]; then grep -n "^E  " gpurun_out/pytest_c8.log | head -20 | cut -c1-300; exit $rc; fi
for v in one cluster sc1 cluster2; do
  if [ "$v" = one ]; then export CCN_FPS_CLUSTER=0; elif [ "$v" = sc1 ]; then export CCN_FPS_CLUSTER=2; else export CCN_FPS_CLUSTER=1; fi
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps 8 --warmup 3 --baseline-config 4 --graph 2>/dev/null | tail -1 > gpurun_out/c8_$v.json || exit 1
  python - $v <<'PY'
import json, sys
d = json.loads(open("gpurun_out/c8_%s.json" % sys.argv[1]).read())
w = d["whole_forward"]
print("%-9s whole-forward graph %.2f ms  eager %.2f ms  bounded eager %.2f ms   feature-pass graph %.1f clouds/s" % (sys.argv[1], w["ms_per_step"], w["eager"]["ms_per_step"], w["bounded_eager_ms"], d["value"]))
PY
done
BENCH_ARGS="--steps 8 --warmup 3 --baseline-config 4 --no-second-line" tools/ab_env.sh "c4one:CCN_FPS_CLUSTER=0" "c4cluster:CCN_FPS_CLUSTER=1"
grep -E " fps" gpurun_out/ab_c4one_kernels.txt gpurun_out/ab_c4cluster_kernels.txt
