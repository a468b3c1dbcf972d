#!/bin/bash
# Round 6, final sources: the whole GPU suite, smoke, the driver's bench command, then `farkle run` end to end (tools/time_farkle_run.py, every
# selector) on tmpfs and on the box's overlay disk, rows mode three times each for the spread.
set -eo pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/ -x -q -m gpu > gpurun_out/r6z_gpu_suite.log 2>&1
tail -3 gpurun_out/r6z_gpu_suite.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r6z_smoke.log 2>&1
tail -1 gpurun_out/r6z_smoke.log
timeout -k 10 300 python bench.py > gpurun_out/r6z_bench.json 2> gpurun_out/r6z_bench.err
cat gpurun_out/r6z_bench.json | cut -c1-400
FK_E2E_DIR=/dev/shm timeout -k 10 300 python tools/time_farkle_run.py 6400 gpurun_out/r6z_e2e_tmpfs.json > gpurun_out/r6z_e2e_tmpfs.log 2>&1
timeout -k 10 300 python tools/time_farkle_run.py 6400 gpurun_out/r6z_e2e_overlay.json > gpurun_out/r6z_e2e_overlay.log 2>&1
for rep in 1 2 3; do
  FK_E2E_DIR=/dev/shm timeout -k 10 200 python tools/time_farkle_run.py 6400 gpurun_out/r6z_rows_tmpfs_$rep.json mega_rows_on,mega_rows_on_v3 > gpurun_out/r6z_rows_tmpfs_$rep.log 2>&1
  timeout -k 10 200 python tools/time_farkle_run.py 6400 gpurun_out/r6z_rows_overlay_$rep.json mega_rows_on,mega_rows_on_v3 > gpurun_out/r6z_rows_overlay_$rep.log 2>&1
done
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/r6z_*e2e*.json") + glob.glob("gpurun_out/r6z_rows_*.json")):
    d = json.load(open(f))
    print(f, {n: round(r["wall_s"], 3) for n, r in d["runs"].items()})
PY
