#!/bin/bash
# Round-3 profile set, part D (after the four-wave hot / cold instances): the PMC passes the traffic stamps of configs 2 and 3 are
# made from (the stamp carries the kernel sources' sha256), SQ / HBM passes of the new k = 5 and k = 6 instances, kernel stats
# of the bench command.  Part E (tools/profile_r03e.sh) prints the bench lines once the stamps exist.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
bash tools/pmc_cfg.sh r03c2 64 2 312500
bash tools/pmc_cfg.sh r03c3 5160 4 77520
bash tools/pmc_cfg.sh r03k6 5160 6 18000
bash tools/pmc_cfg.sh r03k5 5160 5 15000
python3 tools/time_config.py 5160 6 8000 2 0 1 > gpurun_out/r03k6_work.log 2>&1
python3 tools/time_config.py 5160 5 8000 2 0 1 > gpurun_out/r03k5_work.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03_bench_c2_stats -- python3 bench.py --steps 5 --warmup 1 > gpurun_out/r03_bench_c2_under_rocprof.json 2> gpurun_out/r03_bench_c2_under_rocprof.err
echo "c2 stats rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03_bench_c3_stats -- python3 bench.py --config 3 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r03_bench_c3_under_rocprof.json 2> gpurun_out/r03_bench_c3_under_rocprof.err
echo "c3 stats rc=$?"
echo "part D done"
