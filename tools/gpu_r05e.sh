#!/bin/bash
# round 5: buffered half word in the cold-plane slot (16 B of LDS per seat): parity, then kernel times at every player count of the hot / cold kernel
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_hot_cold_gpu.py -x -q > gpurun_out/r5e_pytest_hc.log 2>&1; rc=$?; tail -5 gpurun_out/r5e_pytest_hc.log
[ $rc -ne 0 ] && exit $rc
for k in 5 6 7 8 9 10 11 12; do
  g=5160; n=4000
  if [ $k -eq 9 ] || [ $k -eq 11 ]; then g=5148; fi
  if [ $k -eq 7 ]; then g=5159; fi
  echo "== k=$k hot/cold (auto), grid $g, $n shuffles"; timeout -k 10 200 python tools/time_config.py $g $k $n 4 0 0 2>&1 | tail -3
done > gpurun_out/r5e_time_hc.log 2>&1
cat gpurun_out/r5e_time_hc.log
