#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
echo "== pytest -m gpu" && timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r5i_pytest.log 2>&1; rc=$?; tail -4 gpurun_out/r5i_pytest.log
[ $rc -ne 0 ] && exit $rc
{ echo "# round 5: hot / cold kernels with the LDS image's 32-bit score entries and the additive discard index ('fusedlt') against the nested index ('fused' = the previous commit), alternating processes on one box";
for spec in "5160 5 4000" "5160 6 4000" "5160 8 4000" "5160 10 4000" "5160 12 4000"; do set -- $spec; echo "## grid $1 k $2 shuffles $3"; bash tools/ab_run.sh "python tools/time_config.py $1 $2 $3 4" fused fusedlt 2; done; } > gpurun_out/r5i_ab_fused_lt.log 2>&1
grep -E "^##|^==|play" gpurun_out/r5i_ab_fused_lt.log | awk '/^##|^==/{print; next} {print "   play", $0}' | sed -e 's/grid=.*play \([0-9.]*\) seed.*/\1/' 
