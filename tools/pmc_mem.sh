#!/bin/bash
# usage: tools/pmc_mem.sh <tag> <grid> <k> <n_shuffles> [opt=value ...]
# The vector-memory path of the game kernel: texture-addresser (TA) busy cycles, L1 (TCP) busy / stall cycles and the requests
# it sends to L2, L2 busy.  One rocprofv3 pass per pair of counters of a block (--kernel-trace + --pmc only; the TA / TCP / TCC
# blocks take few counters per pass), every pass under its own timeout.
# Output: gpurun_out/<tag>_mem<i>/ ; summary: python3 tools/pmc_mem_report.py <tag>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=$1; grid=$2; k=$3; nsh=$4; shift 4
run="python3 tools/time_config.py $grid $k $nsh 2 $*"
i=0
for set in "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum" \
           "GRBM_GUI_ACTIVE TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "GRBM_GUI_ACTIVE TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum" \
           "GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" \
           "GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum" \
           "GRBM_GUI_ACTIVE TCC_BUSY_sum TCC_REQ_sum" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" \
           "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"; do
    i=$((i + 1))
    timeout -k 10 150 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/${tag}_mem$i -- $run > gpurun_out/${tag}_mem$i.log 2>&1
    echo "$tag pass $i rc=$?"
done
