#!/bin/bash
# round 6, call 2: FPS fallback tests, the graph tests with bit-identity asserted, index / parallel tests (ABI 4), then the bench with rotating batches
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_index.py tests/test_gpu_graph.py tests/test_abi.py -m gpu -q -x -s --durations=8 > gpurun_out/c2_tests.log 2>&1; rc=$?
tail -n 25 gpurun_out/c2_tests.log; grep -n "fell back" gpurun_out/c2_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-second-line > gpurun_out/r06_vary.json 2> gpurun_out/r06_vary.err; tail -c 600 gpurun_out/r06_vary.err
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-second-line --no-cpu-baseline --vary-batch 1 > gpurun_out/r06_fixed.json 2> gpurun_out/r06_fixed.err
python - <<'PY'
import json
for n in ("vary", "fixed"):
    d = json.loads(open("gpurun_out/r06_%s.json" % n).read().strip().splitlines()[-1])
    print(n, round(d["value"], 2), round(d["ms_per_step"], 2), d["config"].get("batches"), "mallocs", d["config"]["device_mallocs_in_timed_region"],
          "knn", d.get("knn_idx_bit_match"), d.get("knn_checked"), "fallbacks", d["config"].get("fps_cluster_fallbacks"))
PY
