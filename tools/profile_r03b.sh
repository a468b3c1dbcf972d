#!/bin/bash
# Round-3 profile set, part B: vector-memory path counters (TA / TCP / TCC) of the LDS-record kernel at k = 4 and k = 2 and of the
# hot / cold kernel at k = 8 with and without its LDS tables, the four bench lines, rows-mode timing, `farkle run` end to end.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
bash tools/pmc_mem.sh r03_lean4 5160 4 16000 hot_cold=0
bash tools/pmc_mem.sh r03_hc4 5160 4 16000 hot_cold=1
bash tools/pmc_mem.sh r03_hc8 5160 8 24000
bash tools/pmc_mem.sh r03_hc8_global_tables 5160 8 24000 hot_cold_tables=0
bash tools/pmc_mem.sh r03_lean8 5160 8 24000 hot_cold=0
bash tools/pmc_mem.sh r03_c2 64 2 312500
for t in r03_lean4 r03_hc4 r03_hc8 r03_hc8_global_tables r03_lean8 r03_c2; do python3 tools/pmc_mem_report.py $t; done > gpurun_out/r03_pmc_mem_report.txt 2>&1
for c in 2 3 4 5; do timeout -k 10 300 python3 bench.py --config $c > gpurun_out/r03_bench_config$c.json 2> gpurun_out/r03_bench_config$c.err; echo "bench config $c rc=$?"; done
timeout -k 10 200 python3 tools/time_rows.py > gpurun_out/r03_time_rows.log 2>&1
timeout -k 10 200 python3 tools/time_rows.py 1250000 > gpurun_out/r03_time_rows_4e7.log 2>&1
timeout -k 10 300 python3 tools/time_farkle_run.py 51200 gpurun_out/r03_farkle_run_end_to_end.json > gpurun_out/r03_e2e.log 2>&1
timeout -k 10 200 python3 tools/time_h2h_blocks.py 10000 2191 3 > gpurun_out/r03_h2h_blocks.log 2>&1
timeout -k 10 200 python3 tools/exp_hc.py 12000 > gpurun_out/r03_exp_hc.log 2>&1
echo "part B done"
