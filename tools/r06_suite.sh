#!/bin/bash
# the whole -m gpu suite with the parity margins logged (profiles/r06_parity_margins.txt comes from this)
mkdir -p gpurun_out; rm -f gpurun_out/parity_margins.txt
CCN_PARITY_LOG=$PWD/gpurun_out/parity_margins.txt timeout -k 10 1150 python -m pytest tests -m gpu -q --timeout 900 --durations=40 "$@" > gpurun_out/pytest_suite.log 2>&1
rc=$?; tail -n 60 gpurun_out/pytest_suite.log; echo "pytest rc=$rc"; exit $rc
