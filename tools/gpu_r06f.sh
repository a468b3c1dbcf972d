#!/bin/bash
# Round 6, final sources: the whole GPU suite, smoke, the driver's bench command (N = 1).
set -eo pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/ -x -q -m gpu > gpurun_out/r6f_gpu_suite.log 2>&1
tail -3 gpurun_out/r6f_gpu_suite.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r6f_smoke.log 2>&1
tail -1 gpurun_out/r6f_smoke.log
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6f_bench.json 2> gpurun_out/r6f_bench.err
cut -c1-300 gpurun_out/r6f_bench.json
