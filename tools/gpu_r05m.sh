#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r5m_pytest.log 2>&1; rc=$?; tail -3 gpurun_out/r5m_pytest.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python3 tools/time_farkle_run.py 51200 gpurun_out/r5m_farkle_run_end_to_end.json > gpurun_out/r5m_e2e.log 2>&1; echo "e2e rc=$?"
{ for k in 10 12; do for thr in 4 6 8 12 16; do echo "== k=$k batch_threshold=$thr"; timeout -k 10 200 python tools/time_config.py 5160 $k 40000 2 batch_threshold=$thr || exit 1; done; done
} > gpurun_out/r5m_threshold.log 2>&1
grep -E "^==|play" gpurun_out/r5m_threshold.log | sed -e 's/grid=.*device \([0-9.]*\) play \([0-9.]*\) .*/   play \2/'
