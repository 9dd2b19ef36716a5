#!/bin/bash
# round 5: exact FPS over clouds beyond the register-resident form: one 12-byte load per point; tests, then configs[4] forward A/B
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_index.py tests/test_gpu_graph.py -m gpu -q -x --timeout 500 -k "fps or whole_forward or captured" > gpurun_out/pytest_c7.log 2>&1
rc=$?; tail -n 4 gpurun_out/pytest_c7.log | cut -c1-300; echo "pytest rc=$rc"
if [ $rc -ne 0 ]; then grep -n "^E  " gpurun_out/pytest_c7.log | head -20 | cut -c1-300; exit $rc; fi
PREV=$PWD/curvecloudnet_amd/libccn_hip_r05c.so
for v in prev new prev2 new2; do
  if [ "${v#prev}" != "$v" ]; then export CCN_LIB_PATH=$PREV; else unset CCN_LIB_PATH; fi
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps 8 --warmup 3 --baseline-config 4 --graph 2>/dev/null | tail -1 > gpurun_out/c7_$v.json
  python - $v <<'PY'
import json, sys
d = json.loads(open("gpurun_out/c7_%s.json" % sys.argv[1]).read())
w = d["whole_forward"]
print("%-6s whole-forward graph %.2f ms  eager %.2f ms  bounded eager %.2f ms" % (sys.argv[1], w["ms_per_step"], w["eager"]["ms_per_step"], w["bounded_eager_ms"]))
PY
done
unset CCN_LIB_PATH
BENCH_ARGS="--steps 8 --warmup 3 --baseline-config 4 --no-second-line" tools/ab_env.sh "c4prev:CCN_LIB_PATH=$PREV" "c4new:CCN_NOTHING=1"
grep -E " fps" gpurun_out/ab_c4prev_kernels.txt gpurun_out/ab_c4new_kernels.txt
