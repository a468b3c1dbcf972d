#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
echo "== pytest -m gpu" && timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r5h_pytest.log 2>&1; rc=$?; tail -4 gpurun_out/r5h_pytest.log
[ $rc -ne 0 ] && exit $rc
{ echo "# round 5: the discard key assembled from parts that are already in place ('fused': 32-bit score entries carrying the roll's share, the strategy's share = three flag bits) against the decoded-fields form ('before'), alternating processes on one box";
for spec in "64 2 312500" "5160 2 20000" "5160 3 8000" "5160 4 4000"; do set -- $spec; echo "## grid $1 k $2 shuffles $3"; bash tools/ab_run.sh "python tools/time_config.py $1 $2 $3 4" before fused 2; done; } > gpurun_out/r5h_ab_fused_key.log 2>&1
cat gpurun_out/r5h_ab_fused_key.log | grep -v "^grid.*wall 2[0-9]\." 
