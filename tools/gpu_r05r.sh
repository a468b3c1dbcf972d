#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python3 tools/time_farkle_run.py 51200 gpurun_out/r05_farkle_run_end_to_end.json > gpurun_out/r05_e2e.log 2>&1; echo "e2e rc=$?"; grep "^mega" gpurun_out/r05_e2e.log | cut -c1-900
