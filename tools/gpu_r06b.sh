#!/bin/bash
# round 6, after tools/collect_r06.sh: the bench lines again (they now find the round's HBM stamps and instruction mix under profiles/), the missing
# rocprofv3 kernel stats of configs 4 and 5, the 2000-trial fuzz of the shipped plan
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for c in 2 3 4 5 6; do timeout -k 10 400 python3 bench.py --config $c > gpurun_out/r06_bench_config$c.json 2> gpurun_out/r06_bench_config$c.err; echo "bench config $c rc=$?"; done
timeout -k 10 120 python3 bench.py > gpurun_out/r06_bench.json 2> gpurun_out/r06_bench.err; echo "default bench rc=$?"
timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench_driver_command.json 2> gpurun_out/r06_bench_driver_command.err; echo "driver command rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r06_bench_c4_stats -- python3 bench.py --config 4 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/r06_bench_c4_under_rocprof.json 2> gpurun_out/r06_bench_c4_under_rocprof.err; echo "c4 stats rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r06_bench_c5_stats -- python3 bench.py --config 5 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/r06_bench_c5_under_rocprof.json 2> gpurun_out/r06_bench_c5_under_rocprof.err; echo "c5 stats rc=$?"
timeout -k 10 600 python3 tools/fuzz_shipped_plan.py 2000 606 > gpurun_out/r06_fuzz_shipped_plan.log 2>&1; echo "fuzz rc=$?"; tail -3 gpurun_out/r06_fuzz_shipped_plan.log
