#!/bin/bash
# usage: tools/pmc_cfg.sh <tag> <grid> <k> <n_shuffles> [opt=value ...]
# Five rocprofv3 passes over `python3 tools/time_config.py <grid> <k> <n_shuffles> 3`: kernel stats, three SQ counter
# passes, and the HBM passes (FETCH_SIZE / WRITE_SIZE / L2 hit-miss), each --kernel-trace + --pmc only.
# Output: gpurun_out/<tag>_{stats,pmc1..pmc6}/ and gpurun_out/<tag>_*.log
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=$1; grid=$2; k=$3; nsh=$4
shift 4
run="python3 tools/time_config.py $grid $k $nsh 3 $*"
if [ -n "$PMC_RUN" ]; then run="$PMC_RUN"; fi   # another workload under the same passes (e.g. tools/time_h2h_blocks.py)
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats -- $run > gpurun_out/${tag}_stats.log 2>&1
echo "stats done"
timeout -k 10 240 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAVE_CYCLES --output-format csv -d gpurun_out/${tag}_pmc1 -- $run > gpurun_out/${tag}_pmc1.log 2>&1
echo "pmc1 done"
timeout -k 10 240 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS --output-format csv -d gpurun_out/${tag}_pmc2 -- $run > gpurun_out/${tag}_pmc2.log 2>&1
echo "pmc2 done"
timeout -k 10 240 rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_WAVE_CYCLES --output-format csv -d gpurun_out/${tag}_pmc3 -- $run > gpurun_out/${tag}_pmc3.log 2>&1
echo "pmc3 done"
timeout -k 10 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/${tag}_pmc4 -- $run > gpurun_out/${tag}_pmc4.log 2>&1
timeout -k 10 240 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/${tag}_pmc5 -- $run > gpurun_out/${tag}_pmc5.log 2>&1
timeout -k 10 240 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/${tag}_pmc6 -- $run > gpurun_out/${tag}_pmc6.log 2>&1
echo "hbm done"
