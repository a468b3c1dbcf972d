#!/bin/bash
# usage: tools/pmc_quick.sh <tag> <grid> <k> <n_shuffles>   — two SQ passes only (instruction counts, lane utilisation)
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=$1; grid=$2; k=$3; nsh=$4
run="python3 tools/time_config.py $grid $k $nsh 3"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAVE_CYCLES --output-format csv -d gpurun_out/${tag}_pmc1 -- $run > gpurun_out/${tag}_pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_WAVE_CYCLES --output-format csv -d gpurun_out/${tag}_pmc3 -- $run > gpurun_out/${tag}_pmc3.log 2>&1
