#!/bin/bash
# Round-3 profile set, part A (run on the GPU box): kernel stats of the bench command (configs 2 and 3), SQ / HBM PMC passes for
# config 2, config 3 and k = 8 (hot / cold kernel), per-game work logs.  Every rocprofv3 run under its own timeout.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03_bench_c2_stats -- python3 bench.py --steps 5 --warmup 1 > gpurun_out/r03_bench_c2_under_rocprof.json 2> gpurun_out/r03_bench_c2_under_rocprof.err
echo "c2 stats rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03_bench_c3_stats -- python3 bench.py --config 3 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r03_bench_c3_under_rocprof.json 2> gpurun_out/r03_bench_c3_under_rocprof.err
echo "c3 stats rc=$?"
bash tools/pmc_cfg.sh r03c2 64 2 312500
bash tools/pmc_cfg.sh r03c3 5160 4 77520
bash tools/pmc_cfg.sh r03k8 5160 8 24000
python3 tools/time_config.py 64 2 312500 2 42 1 > gpurun_out/r03c2_work.log 2>&1
python3 tools/time_config.py 5160 4 8000 2 0 1 > gpurun_out/r03c3_work.log 2>&1
python3 tools/time_config.py 5160 8 8000 2 0 1 > gpurun_out/r03k8_work.log 2>&1
echo "part A done"
