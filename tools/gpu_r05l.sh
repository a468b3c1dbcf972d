#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_hot_cold_gpu.py -x -q > gpurun_out/r5l_pytest.log 2>&1; rc=$?; tail -5 gpurun_out/r5l_pytest.log
[ $rc -ne 0 ] && exit $rc
{ for k in 5 6 8 10 12; do
    for ip in 0 1; do echo "== k=$k hot_cold_ip=$ip"; timeout -k 10 200 python tools/time_config.py 5160 $k 40000 3 hot_cold_ip=$ip clock_stamps=1 || exit 1; done
  done
  echo "== k=8 hot_cold_ip=2 (three waves)"; timeout -k 10 200 python tools/time_config.py 5160 8 40000 3 hot_cold_ip=2 clock_stamps=1
} > gpurun_out/r5l_ip.log 2>&1
cut -c1-40,70-260 gpurun_out/r5l_ip.log
