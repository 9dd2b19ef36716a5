#!/bin/bash
# round 5: cluster FPS in the models that use it (A2D2 sections, graph capture), then the configs[4] lines
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_index.py tests/test_gpu_graph.py tests/test_gpu_golden.py tests/test_gpu_model.py -m gpu -q -x --timeout 600 -k "fps or graph or captured or a2d2 or golden or reference_vectors or mixed" > gpurun_out/pytest_c9.log 2>&1
rc=$?; tail -n 4 gpurun_out/pytest_c9.log | cut -c1-300; echo "pytest rc=$rc"
if [ $rc -ne 0 ]; then grep -n "^E  " gpurun_out/pytest_c9.log | head -20 | cut -c1-300; exit $rc; fi
