#!/bin/bash
# Round 6: the A/B evidence of the rows-mode work, final sources — async rows calls (engine only), page-locked allocation routes, the phase
# trace of the production sweep.
set -eo pipefail
mkdir -p gpurun_out
timeout -k 10 200 python tools/time_rows_async.py > gpurun_out/r6y_rows_async.txt 2>&1
timeout -k 10 100 python tools/pinned_alloc_bench.py > gpurun_out/r6y_pinned_alloc.txt 2>&1
FK_RUN_TRACE=1 FK_SHARD_WRITER_TIMING=1 FK_E2E_DIR=/dev/shm timeout -k 10 200 python tools/time_farkle_run.py 6400 gpurun_out/r6y_trace.json mega_rows_on,mega_rows_on_v3 > gpurun_out/r6y_trace.log 2> gpurun_out/r6y_trace.err
python tools/trace_summary.py gpurun_out/r6y_trace.err > gpurun_out/r6y_trace_summary.txt
for mode in "FK_ROWS_ASYNC=0" "FK_ROWS_SLOTS=2" "FK_ROWS_PIPELINE=1" "FK_ROWS_ASYNC=1"; do
  env $mode FK_E2E_DIR=/dev/shm timeout -k 10 200 python tools/time_farkle_run.py 6400 gpurun_out/r6y_ab.json mega_rows_on,mega_rows_on_v3 > gpurun_out/r6y_ab.log 2>&1
  python - <<PY >> gpurun_out/r6y_switches.txt
import json
d = json.load(open("gpurun_out/r6y_ab.json"))
print("$mode", {n: round(r["wall_s"], 3) for n, r in d["runs"].items()}, "engine_s", {n: round(r["engine_s"], 3) for n, r in d["runs"].items()})
PY
done
cat gpurun_out/r6y_switches.txt
