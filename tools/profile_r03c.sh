#!/bin/bash
# Round-3 profile set, part C: SQ / HBM PMC passes for the two kernels that had none — the hot / cold kernel at k = 6 and the
# batched-H2H instance (10 000 production-size blocks) — and the final bench lines with the traffic stamps attached.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
bash tools/pmc_cfg.sh r03k6 5160 6 18000
PMC_RUN="python3 tools/time_h2h_blocks.py 10000 2191 2" bash tools/pmc_cfg.sh r03h2h 0 0 0
python3 tools/time_config.py 5160 6 8000 2 0 1 > gpurun_out/r03k6_work.log 2>&1
for c in 2 3 4 5; do timeout -k 10 300 python3 bench.py --config $c > gpurun_out/r03_bench_config$c.json 2> gpurun_out/r03_bench_config$c.err; echo "bench config $c rc=$?"; done
timeout -k 10 120 python3 bench.py > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err; echo "default bench rc=$?"
echo "part C done"
