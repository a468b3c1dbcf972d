#!/bin/bash
# round 4, first GPU profile call: the k = 8 hot / cold kernel with the cold records in REGISTERS (option hot_cold_cold_regs=2)
# under the same PMC passes as round 3's plane instance, plus the L2 write-path counters of both instances.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --list-avail > gpurun_out/r4_counters_avail.txt 2>&1
bash tools/pmc_cfg.sh r4k8cr 5160 8 24000 hot_cold_cold_regs=2 &&
for v in 0 2; do
  run="python3 tools/time_config.py 5160 8 24000 3 hot_cold_cold_regs=$v"
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_WRITE_sum TCC_REQ_sum --output-format csv -d gpurun_out/r4k8w${v}_a -- $run > gpurun_out/r4k8w${v}_a.log 2>&1
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc TCC_NORMAL_WRITEBACK_sum TCC_NORMAL_EVICT_sum TCC_ALL_TC_OP_WB_WRITEBACK_sum --output-format csv -d gpurun_out/r4k8w${v}_b -- $run > gpurun_out/r4k8w${v}_b.log 2>&1
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc MemUnitStalled VALUBusy MeanOccupancyPerActiveCU --output-format csv -d gpurun_out/r4k8w${v}_c -- $run > gpurun_out/r4k8w${v}_c.log 2>&1
  echo "variant $v done"
done
