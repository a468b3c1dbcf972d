#!/bin/bash
# Round-4 profile set, part B (after the host-side change that lets the next chunk's permutation kernels run beside the current
# chunk's seeding tail; kernel sources unchanged, so the PMC passes and traffic stamps of part A stand): bench lines, kernel stats
# of the bench commands, `farkle run` end to end, the two-rank rehearsals.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04_bench_c2_stats -- python3 bench.py --steps 5 --warmup 1 > gpurun_out/r04_bench_c2_under_rocprof.json 2> gpurun_out/r04_bench_c2_under_rocprof.err
echo "c2 stats rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04_bench_c3_stats -- python3 bench.py --config 3 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r04_bench_c3_under_rocprof.json 2> gpurun_out/r04_bench_c3_under_rocprof.err
echo "c3 stats rc=$?"
for c in 2 3 4 5; do timeout -k 10 300 python3 bench.py --config $c > gpurun_out/r04_bench_config$c.json 2> gpurun_out/r04_bench_config$c.err; echo "bench config $c rc=$?"; done
timeout -k 10 120 python3 bench.py > gpurun_out/r04_bench.json 2> gpurun_out/r04_bench.err; echo "default bench rc=$?"
timeout -k 10 300 python3 tools/time_farkle_run.py 51200 gpurun_out/r04_farkle_run_end_to_end.json > gpurun_out/r04_e2e.log 2>&1; echo "e2e rc=$?"
FK_DIST_BACKEND=gloo timeout -k 10 200 python3 bench.py --gpus 2 --steps 3 --warmup 1 > gpurun_out/r04_bench_2rank_gloo_one_gpu.json 2> gpurun_out/r04_bench_2rank_gloo.err; echo "2-rank gloo rc=$?"
FK_BENCH_SHARE_GPU=1 timeout -k 10 200 python3 bench.py --gpus 2 --steps 3 --warmup 1 > gpurun_out/r04_bench_2rank_one_gpu_rccl_refused_fallback.json 2> gpurun_out/r04_bench_2rank_rccl.err; echo "2-rank rccl-refused rc=$?"
echo "profile r04b done"
