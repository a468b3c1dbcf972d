#!/bin/bash
# usage: tools/pmc_seed.sh <tag> <grid> <k> <n_shuffles>  — counter passes for the preparation kernels (report with
# `python tools/pmc_report.py <tag> <games> fk_seed`); FARKLE_HIP_LIB selects the library build
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=$1; grid=$2; k=$3; nsh=$4
run="python3 tools/time_config.py $grid $k $nsh 3"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAVE_CYCLES --output-format csv -d gpurun_out/${tag}_pmc1 -- $run > gpurun_out/${tag}_pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS --output-format csv -d gpurun_out/${tag}_pmc2 -- $run > gpurun_out/${tag}_pmc2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/${tag}_pmc4 -- $run > gpurun_out/${tag}_pmc4.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/${tag}_pmc5 -- $run > gpurun_out/${tag}_pmc5.log 2>&1
