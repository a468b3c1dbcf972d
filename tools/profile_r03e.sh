#!/bin/bash
# Round-3 profile set, part E: the final bench lines (traffic stamps attached), `farkle run` end to end, the variants log.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for c in 2 3 4 5; do timeout -k 10 300 python3 bench.py --config $c > gpurun_out/r03_bench_config$c.json 2> gpurun_out/r03_bench_config$c.err; echo "bench config $c rc=$?"; done
timeout -k 10 120 python3 bench.py > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err; echo "default bench rc=$?"
timeout -k 10 300 python3 tools/time_farkle_run.py 51200 gpurun_out/r03_farkle_run_end_to_end.json > gpurun_out/r03_e2e.log 2>&1; echo "e2e rc=$?"
echo "part E done"
