#!/bin/bash
# Round 6: the async-rows A/B (FK_ROWS_ASYNC=0 / 1) of the production sweep with rows on, on tmpfs, after the GPU tests of the path.
set -eo pipefail
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_shard_writer.py tests/test_contract_v3.py tests/test_host_gpu.py -x -q -m gpu > gpurun_out/r6i_tests.log 2>&1
tail -3 gpurun_out/r6i_tests.log
for mode in 0 1 0 1; do
  FK_ROWS_ASYNC=$mode FK_E2E_DIR=/dev/shm timeout -k 10 200 python tools/time_farkle_run.py 6400 gpurun_out/r6i_e2e_async${mode}.json mega_rows_on,mega_rows_on_v3 > gpurun_out/r6i_e2e_async${mode}.log 2>&1
  python - <<PY
import json
d = json.load(open("gpurun_out/r6i_e2e_async${mode}.json"))
for n, r in d["runs"].items():
    if isinstance(r, dict) and "wall_s" in r:
        print("async=${mode}", n, "wall", r["wall_s"], "engine", r.get("engine_s"), "shard", r.get("shard_s"))
PY
done
