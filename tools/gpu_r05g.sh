#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
echo "== pytest -m gpu" && timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r5g_pytest.log 2>&1; rc=$?; tail -4 gpurun_out/r5g_pytest.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 500 python3 tools/time_farkle_run.py 51200 gpurun_out/r5g_farkle_run_end_to_end.json > gpurun_out/r5g_e2e.log 2>&1; echo "e2e rc=$?"; grep -E "^(rows_off|rows_off_metric_chunks|rows_on|config3)" gpurun_out/r5g_e2e.log | cut -c1-400
