#!/bin/bash
# quick GPU check: parity tests, then the micro-benchmarks given in $KB (default k1,k2)
set -u
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -n 8 gpurun_out/pytest_gpu.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 400 python tools/kbench.py --rows ${ROWS:-1000000} --only ${KB:-k1,k2} ${KBARGS:-} > gpurun_out/kbench.log 2>&1; rc=$?
echo "kbench rc=$rc"; cat gpurun_out/kbench.log | cut -c1-400
