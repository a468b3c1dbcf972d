#!/bin/bash
# Round-2 profile set (run on the GPU box): kernel stats of the bench command (configs 2 and 3), PMC passes, plain bench lines.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02_bench_c2_stats -- python3 bench.py --steps 5 --warmup 1 > gpurun_out/r02_bench_c2_under_rocprof.json 2> gpurun_out/r02_bench_c2_under_rocprof.err
echo "c2 stats done"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02_bench_c3_stats -- python3 bench.py --config 3 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r02_bench_c3_under_rocprof.json 2> gpurun_out/r02_bench_c3_under_rocprof.err
echo "c3 stats done"
bash tools/pmc_cfg.sh r02c2 64 2 312500
bash tools/pmc_cfg.sh r02c3 5160 4 77520
python3 tools/time_config.py 64 2 312500 2 42 1 > gpurun_out/r02c2_work.log 2>&1
python3 tools/time_config.py 5160 4 8000 2 0 1 > gpurun_out/r02c3_work.log 2>&1
for c in 2 3 4 5; do python3 bench.py --config $c > gpurun_out/r02_bench_config$c.json 2> gpurun_out/r02_bench_config$c.err; echo "bench config $c rc=$?"; done
python3 tools/time_h2h_blocks.py 10000 2191 3 > gpurun_out/r02_h2h_blocks.log 2>&1
