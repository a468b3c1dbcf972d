#!/bin/bash
# Round 6: rows mode of the production sweep end to end (tmpfs) after the GPU tests of the path; FK_ROWS_PIPELINE A/B; the last run carries
# FK_RUN_TRACE / FK_SHARD_WRITER_TIMING (diagnostics: phase trace of the host threads, writer thread time).
set -eo pipefail
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_shard_writer.py tests/test_contract_v3.py tests/test_host_gpu.py tests/test_runner.py -x -q -m gpu > gpurun_out/r6m_tests.log 2>&1
tail -3 gpurun_out/r6m_tests.log
for mode in 1 0 1 0; do
  FK_ROWS_PIPELINE=$mode FK_E2E_DIR=/dev/shm timeout -k 10 200 python tools/time_farkle_run.py 6400 gpurun_out/r6m_e2e_p$mode.json mega_rows_on,mega_rows_on_v3 > gpurun_out/r6m_e2e_p$mode.log 2>&1
  python - <<PY
import json
d = json.load(open("gpurun_out/r6m_e2e_p$mode.json"))
for n, r in d["runs"].items():
    print("pipeline $mode", n, "wall %.3f" % r["wall_s"], "engine %.3f" % r["engine_s"], "writer %.3f" % r["row_shard_write_s"], "bytes %.2f GB" % (r["row_shard_bytes"] / 1e9))
PY
done
FK_RUN_TRACE=1 FK_SHARD_WRITER_TIMING=1 FK_E2E_DIR=/dev/shm timeout -k 10 200 python tools/time_farkle_run.py 6400 gpurun_out/r6m_trace.json mega_rows_on,mega_rows_on_v3 > gpurun_out/r6m_trace.log 2> gpurun_out/r6m_trace.err
