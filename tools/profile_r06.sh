#!/bin/bash
# Round-6 profile set on the FINAL kernel sources (traffic stamps are keyed by their sha256): PMC passes + HBM-traffic stamps of configs 2, 3, 5
# and of every player count of the production sweep (bench --config 6: k = 2, 3, 4, 5, 6, 8, 10, 12 at 10^8 games each; config 4's
# k = 2, 4, 6, 8 at its own launch sizes), PMC summaries per player count, kernel stats of the bench commands, the bench lines,
# `farkle run` end to end (configs 2 and 3), the two-rank rehearsals on the one GPU.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
R=6
part=${1:-all}
if [ "$part" = all ] || [ "$part" = a ] || [ "$part" = a1 ]; then
bash tools/pmc_cfg.sh r06c2 64 2 312500 && python3 tools/make_traffic_json.py r06c2 2 760000000 "tools/pmc_cfg.sh r06c2 64 2 312500 (tools/profile_r06.sh)" $R
python3 tools/pmc_report.py r06c2 1446000000 fk_play > gpurun_out/r06_play_kernel_pmc_summary_config2.txt 2>&1
bash tools/pmc_cfg.sh r06c3 5160 4 77520 && python3 tools/make_traffic_json.py r06c3 3 17600000000 "tools/pmc_cfg.sh r06c3 5160 4 77520 (tools/profile_r06.sh)" $R
python3 tools/pmc_report.py r06c3 22990000000 fk_play > gpurun_out/r06_play_hc_kernel_pmc_summary_config3.txt 2>&1
echo "configs 2, 3 done"
fi
if [ "$part" = all ] || [ "$part" = a ] || [ "$part" = a1 ] || [ "$part" = a2 ]; then
SPECS=("2 38759 154.0" "3 58139 187.4" "4 77519 229.9" "5 96899 273.3" "6 116279 316.0" "8 155038 399.2" "10 193798 479.5" "12 232558 557.5")
[ "$part" = a1 ] && SPECS=("2 38759 154.0" "3 58139 187.4" "4 77519 229.9" "5 96899 273.3")
[ "$part" = a2 ] && SPECS=("6 116279 316.0" "8 155038 399.2" "10 193798 479.5" "12 232558 557.5")
# config 6: one stamp + one PMC summary per player count at the sweep's launch size (10^8 games per k; algorithmic bytes = (34 k + 40) per game)
for spec in "${SPECS[@]}"; do
  set -- $spec; k=$1; nsh=$2; rolls=$3
  games=$(( nsh * (5160 / k) ))
  bash tools/pmc_cfg.sh r06s$k 5160 $k $nsh && python3 tools/make_traffic_json.py r06s$k 6 $(( games * (34 * k + 40) )) "tools/pmc_cfg.sh r06s$k 5160 $k $nsh" $R $k
  python3 tools/pmc_report.py r06s$k $(python3 -c "print($games * $rolls)") fk_play > gpurun_out/r06_pmc_summary_config6_k$k.txt 2>&1
  echo "config 6 k=$k done"
done
fi
if [ "$part" = all ] || [ "$part" = b ]; then
# config 4's own launch sizes (2.5 x 10^8 games per k; k >= 6 in two launches)
bash tools/pmc_cfg.sh r06k2 5160 2 96899 && python3 tools/make_traffic_json.py r06k2 4 27000000000 "tools/pmc_cfg.sh r06k2 5160 2 96899" $R 2
bash tools/pmc_cfg.sh r06k4 5160 4 193798 && python3 tools/make_traffic_json.py r06k4 4 44000000000 "tools/pmc_cfg.sh r06k4 5160 4 193798" $R 4
bash tools/pmc_cfg.sh r06k6 5160 6 145348 && python3 tools/make_traffic_json.py r06k6 4 30500000000 "tools/pmc_cfg.sh r06k6 5160 6 145348 (one of the k = 6 call's two launches)" $R 6
bash tools/pmc_cfg.sh r06k8 5160 8 193798 && python3 tools/make_traffic_json.py r06k8 4 39000000000 "tools/pmc_cfg.sh r06k8 5160 8 193798 (one of the k = 8 call's two launches)" $R 8
echo "config 4 stamps done"
PMC_RUN="python3 bench.py --config 5 --steps 1 --warmup 0 --no-cpu-baseline" bash tools/pmc_cfg.sh r06c5 0 0 0 && python3 tools/make_traffic_json.py r06c5 5 45600000000 "bench.py --config 5 --steps 1 --warmup 0 under tools/pmc_cfg.sh (6 x 10^8 attempts per launch x 76 B)" $R
echo "config 5 stamp done"
fi
if [ "$part" = all ] || [ "$part" = c ]; then
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r06_bench_c2_stats -- python3 bench.py --steps 5 --warmup 1 > gpurun_out/r06_bench_c2_under_rocprof.json 2> gpurun_out/r06_bench_c2_under_rocprof.err
echo "c2 stats rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r06_bench_c3_stats -- python3 bench.py --config 3 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r06_bench_c3_under_rocprof.json 2> gpurun_out/r06_bench_c3_under_rocprof.err
echo "c3 stats rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r06_bench_c6_stats -- python3 bench.py --config 6 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/r06_bench_c6_under_rocprof.json 2> gpurun_out/r06_bench_c6_under_rocprof.err
echo "c6 stats rc=$?"
for c in 2 3; do timeout -k 10 400 python3 bench.py --config $c > gpurun_out/r06_bench_config$c.json 2> gpurun_out/r06_bench_config$c.err; echo "bench config $c rc=$?"; done
fi
if [ "$part" = all ] || [ "$part" = d ]; then
for c in 4 5 6; do timeout -k 10 400 python3 bench.py --config $c > gpurun_out/r06_bench_config$c.json 2> gpurun_out/r06_bench_config$c.err; echo "bench config $c rc=$?"; done
timeout -k 10 120 python3 bench.py > gpurun_out/r06_bench.json 2> gpurun_out/r06_bench.err; echo "default bench rc=$?"
FK_E2E_DIR=/dev/shm timeout -k 10 600 python3 tools/time_farkle_run.py 51200 gpurun_out/r06_farkle_run_end_to_end.json > gpurun_out/r06_e2e.log 2>&1; echo "e2e (tmpfs) rc=$?"
timeout -k 10 600 python3 tools/time_farkle_run.py 51200 gpurun_out/r06_farkle_run_end_to_end_overlay_disk.json rows_on,mega_rows_on,mega_rows_on_v3 > gpurun_out/r06_e2e_disk.log 2>&1; echo "e2e (overlay disk) rc=$?"
# the clock stamps' cost inside the timed steps (advisor, round 5): the same command with and without them
FK_BENCH_CLOCK_STAMPS=0 timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r06_bench_clock_stamps_off.json 2> gpurun_out/r06_bench_clock_stamps_off.err; echo "stamps off rc=$?"
timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r06_bench_clock_stamps_on.json 2> gpurun_out/r06_bench_clock_stamps_on.err; echo "stamps on rc=$?"
timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench_driver_command.json 2> gpurun_out/r06_bench_driver_command.err; echo "driver command rc=$?"
FK_DIST_BACKEND=gloo timeout -k 10 400 python3 bench.py --gpus 4 --config 5 --steps 1 --warmup 1 > gpurun_out/r06_bench_4rank_gloo_one_gpu_config5.json 2> gpurun_out/r06_bench_4rank_gloo_config5.err; echo "4-rank gloo config 5 rc=$?"
FK_DIST_BACKEND=gloo timeout -k 10 200 python3 bench.py --gpus 2 --steps 3 --warmup 1 > gpurun_out/r06_bench_2rank_gloo_one_gpu.json 2> gpurun_out/r06_bench_2rank_gloo.err; echo "2-rank gloo rc=$?"
FK_BENCH_SHARE_GPU=1 timeout -k 10 200 python3 bench.py --gpus 2 --steps 3 --warmup 1 > gpurun_out/r06_bench_2rank_one_gpu_rccl_refused_fallback.json 2> gpurun_out/r06_bench_2rank_rccl.err; echo "2-rank rccl-refused rc=$?"
fi
echo "profile r06 part $part done"
