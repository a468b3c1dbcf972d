#!/bin/bash
# Round-4 profile set on the final kernel sources: PMC passes + HBM-traffic stamps of every bench workload (config 4 per player
# count, config 5 under the bench command itself), kernel stats of the bench commands, the bench lines, `farkle run` end to end,
# the lag post-pass timing, the two-rank rehearsals on the one GPU.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
R=4
bash tools/pmc_cfg.sh r04c2 64 2 312500 && python3 tools/make_traffic_json.py r04c2 2 760000000 "tools/pmc_cfg.sh r04c2 64 2 312500 (tools/profile_r04.sh)" $R
bash tools/pmc_cfg.sh r04c3 5160 4 77520 && python3 tools/make_traffic_json.py r04c3 3 17600000000 "tools/pmc_cfg.sh r04c3 5160 4 77520 (tools/profile_r04.sh)" $R
# config 4: one stamp per player count at the sweep's launch sizes (2.5 x 10^8 games per k; algorithmic bytes = (34 k + 40) per game)
bash tools/pmc_cfg.sh r04k2 5160 2 96899 && python3 tools/make_traffic_json.py r04k2 4 27000000000 "tools/pmc_cfg.sh r04k2 5160 2 96899" $R 2
bash tools/pmc_cfg.sh r04k4 5160 4 193798 && python3 tools/make_traffic_json.py r04k4 4 44000000000 "tools/pmc_cfg.sh r04k4 5160 4 193798" $R 4
bash tools/pmc_cfg.sh r04k6 5160 6 145348 && python3 tools/make_traffic_json.py r04k6 4 30500000000 "tools/pmc_cfg.sh r04k6 5160 6 145348 (one of the k = 6 call's two launches)" $R 6
bash tools/pmc_cfg.sh r04k8 5160 8 193798 && python3 tools/make_traffic_json.py r04k8 4 39000000000 "tools/pmc_cfg.sh r04k8 5160 8 193798 (one of the k = 8 call's two launches)" $R 8
echo "config 2-4 stamps done"
PMC_RUN="python3 bench.py --config 5 --steps 1 --warmup 0 --no-cpu-baseline" bash tools/pmc_cfg.sh r04c5 0 0 0 && python3 tools/make_traffic_json.py r04c5 5 45600000000 "bench.py --config 5 --steps 1 --warmup 0 under tools/pmc_cfg.sh (6 x 10^8 attempts per launch x 76 B)" $R
echo "config 5 stamp done"
for k in 2 8; do python3 tools/pmc_report.py r04k$k 1 fk_play > gpurun_out/r04_pmc_summary_k$k.txt 2>&1; done
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04_bench_c2_stats -- python3 bench.py --steps 5 --warmup 1 > gpurun_out/r04_bench_c2_under_rocprof.json 2> gpurun_out/r04_bench_c2_under_rocprof.err
echo "c2 stats rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04_bench_c3_stats -- python3 bench.py --config 3 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r04_bench_c3_under_rocprof.json 2> gpurun_out/r04_bench_c3_under_rocprof.err
echo "c3 stats rc=$?"
for c in 2 3 4 5; do timeout -k 10 300 python3 bench.py --config $c > gpurun_out/r04_bench_config$c.json 2> gpurun_out/r04_bench_config$c.err; echo "bench config $c rc=$?"; done
timeout -k 10 120 python3 bench.py > gpurun_out/r04_bench.json 2> gpurun_out/r04_bench.err; echo "default bench rc=$?"
timeout -k 10 300 python3 tools/time_farkle_run.py 51200 gpurun_out/r04_farkle_run_end_to_end.json > gpurun_out/r04_e2e.log 2>&1; echo "e2e rc=$?"
timeout -k 10 200 python3 tools/time_lags.py > gpurun_out/r04_lag_post_pass_timing.log 2>&1; echo "lags rc=$?"
FK_DIST_BACKEND=gloo timeout -k 10 200 python3 bench.py --gpus 2 --steps 3 --warmup 1 > gpurun_out/r04_bench_2rank_gloo_one_gpu.json 2> gpurun_out/r04_bench_2rank_gloo.err; echo "2-rank gloo rc=$?"
FK_BENCH_SHARE_GPU=1 timeout -k 10 200 python3 bench.py --gpus 2 --steps 3 --warmup 1 > gpurun_out/r04_bench_2rank_one_gpu_rccl_refused_fallback.json 2> gpurun_out/r04_bench_2rank_rccl.err; echo "2-rank rccl-refused rc=$?"
echo "profile r04 done"
