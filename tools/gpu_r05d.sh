#!/bin/bash
# round 5: nine-to-twelve-seat instances of the hot / cold kernel — parity, then time against the LDS-record kernel (hot_cold = 0) on the 5 160 grid
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_hot_cold_gpu.py -x -q > gpurun_out/r5d_pytest_hc.log 2>&1; rc=$?; tail -5 gpurun_out/r5d_pytest_hc.log
[ $rc -ne 0 ] && exit $rc
for k in 9 10 11 12; do
  n=$((3000)); 
  g=5160; if [ $k -eq 9 ] || [ $k -eq 11 ]; then g=5148; fi
  echo "== k=$k hot/cold (auto)"; timeout -k 10 200 python tools/time_config.py $g $k $n 3 0 1 2>&1 | tail -3
  echo "== k=$k LDS records (hot_cold=0)"; timeout -k 10 200 python tools/time_config.py $g $k $n 3 0 0 hot_cold=0 2>&1 | tail -2
done > gpurun_out/r5d_time_wide.log 2>&1
cat gpurun_out/r5d_time_wide.log
