#!/bin/bash
# A/B of whole-step variants selected by environment variables, one box, back to back:
#   tools/ab_env.sh "NAME1:VAR=val VAR2=val" "NAME2:..."     (bench arguments via BENCH_ARGS)
mkdir -p gpurun_out
for spec in "$@"; do
  name="${spec%%:*}"; envs="${spec#*:}"
  env $envs timeout -k 10 300 python bench.py --no-cpu-baseline ${BENCH_ARGS:---steps 16} > gpurun_out/ab_$name.json 2> gpurun_out/ab_$name.err || echo "$name failed"
  cp gpurun_out/bench_kernels.txt gpurun_out/ab_${name}_kernels.txt 2>/dev/null
  cp gpurun_out/bench_gemm_shapes.txt gpurun_out/ab_${name}_shapes.txt 2>/dev/null
  python - "$name" <<'PY'
import json, sys
name = sys.argv[1]
try:
    d = json.loads(open("gpurun_out/ab_%s.json" % name).read().strip().splitlines()[-1])
    r = d.get("roofline", {})
    print("%-14s %7.2f clouds/s  %7.2f ms/step  dominant %.1f (TFLOP/s, or GB/s where HBM-bound)  all-GEMM %.1f TFLOP/s busy %.1f ms  kernel time %.1f ms"
          % (name, d["value"], d["ms_per_step"], r.get("achieved") or 0, r.get("all_gemm_launches", {}).get("achieved", 0),
             r.get("all_gemm_launches", {}).get("busy_ms_per_step", 0), d.get("kernel_time_ms_per_step", 0)))
except Exception as e:
    print(name, "no result:", e)
PY
done
