#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r5n_pytest.log 2>&1; rc=$?; tail -3 gpurun_out/r5n_pytest.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python bench.py --config 6 --no-cpu-baseline > gpurun_out/r5n_bench6.json 2> gpurun_out/r5n_bench6.err; echo "bench6 rc=$?"
