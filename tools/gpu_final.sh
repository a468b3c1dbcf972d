#!/bin/bash
# the driver's round-end sequence, rehearsed: build check, GPU suite, smoke, default bench
set -o pipefail
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" || exit 1
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/final_pytest.log 2>&1; rc=$?; tail -3 gpurun_out/final_pytest.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" || exit 1
timeout -k 10 300 python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err; echo "bench rc=$?"; cut -c1-400 gpurun_out/final_bench.json
