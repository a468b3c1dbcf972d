#!/bin/bash
# round 6: memory-pressure tests, the four-rank one-GPU rehearsal of config 5 that died of hipErrorOutOfMemory in round 5, the full GPU suite
set -o pipefail
mkdir -p gpurun_out
echo "== memory tests" && timeout -k 10 300 python -m pytest tests/test_memory_gpu.py -x -q -m gpu > gpurun_out/r6k_memory.log 2>&1; rc=$?; tail -2 gpurun_out/r6k_memory.log
[ $rc -ne 0 ] && exit $rc
echo "== 4 ranks on one GPU, gloo, config 5" && FK_DIST_BACKEND=gloo timeout -k 10 400 python3 bench.py --gpus 4 --config 5 --steps 1 --warmup 1 > gpurun_out/r6k_bench_4rank_gloo_one_gpu_config5.json 2> gpurun_out/r6k_bench_4rank_gloo_config5.err; rc=$?; echo "rc=$rc"; tail -c 600 gpurun_out/r6k_bench_4rank_gloo_one_gpu_config5.json
[ $rc -ne 0 ] && { tail -5 gpurun_out/r6k_bench_4rank_gloo_config5.err; exit $rc; }
echo "== pytest -m gpu" && timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r6k_pytest.log 2>&1; rc=$?; tail -3 gpurun_out/r6k_pytest.log
exit $rc
