#!/bin/bash
# round 5: DPP arg-max in the FPS kernels: bit-identity tests, configs[4] forward and kernel tables A/B against the previous library
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_index.py tests/test_gpu_graph.py tests/test_gpu_golden.py -m gpu -q -x --timeout 500 > gpurun_out/pytest_c10.log 2>&1
rc=$?; tail -n 4 gpurun_out/pytest_c10.log | cut -c1-300; echo "pytest rc=$rc"
if [ $rc -ne 0 ]; then grep -n "^E  " gpurun_out/pytest_c10.log | head -20 | cut -c1-300; exit $rc; fi
PREV=$PWD/curvecloudnet_amd/libccn_hip_r05d.so
for v in prev new prev2 new2; do
  if [ "${v#prev}" != "$v" ]; then export CCN_LIB_PATH=$PREV; else unset CCN_LIB_PATH; fi
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps 8 --warmup 3 --baseline-config 4 --graph 2>/dev/null | tail -1 > gpurun_out/c10_$v.json || exit 1
  python - $v <<'PY'
import json, sys
d = json.loads(open("gpurun_out/c10_%s.json" % sys.argv[1]).read())
w = d["whole_forward"]
print("%-9s whole-forward graph %.2f ms  eager %.2f ms  bounded eager %.2f ms   feature-pass graph %.1f clouds/s" % (sys.argv[1], w["ms_per_step"], w["eager"]["ms_per_step"], w["bounded_eager_ms"], d["value"]))
PY
done
unset CCN_LIB_PATH
BENCH_ARGS="--steps 8 --warmup 3 --baseline-config 4 --no-second-line" tools/ab_env.sh "c4prev:CCN_LIB_PATH=$PREV" "c4new:CCN_NOTHING=1"
BENCH_ARGS="--steps 12 --no-second-line" tools/ab_env.sh "kprev:CCN_LIB_PATH=$PREV" "knew:CCN_NOTHING=1"
grep -E " fps" gpurun_out/ab_c4prev_kernels.txt gpurun_out/ab_c4new_kernels.txt gpurun_out/ab_kprev_kernels.txt gpurun_out/ab_knew_kernels.txt
