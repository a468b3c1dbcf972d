#!/bin/bash
# usage: tools/pmc_extra.sh <tag> <grid> <k> <n_shuffles> [opt=value ...]
# Extra PMC passes (one rocprofv3 run each, --kernel-trace + --pmc only): instruction cache, instruction-fetch / vector-memory / LDS
# latency, occupancy and unit-busy figures.  Output: gpurun_out/<tag>_x{1..5}/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=$1; grid=$2; k=$3; nsh=$4
shift 4
run="python3 tools/time_config.py $grid $k $nsh 3 $*"
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "InstrFetchLatency" "VmemLatency" "MeanOccupancyPerActiveCU VALUBusy SALUBusy" "MemUnitStalled LdsLatency"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/${tag}_x$i -- $run > gpurun_out/${tag}_x$i.log 2>&1
  echo "$tag pass $i rc=$?"
done
