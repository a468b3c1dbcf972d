#!/bin/bash
# Round 6: turn the counter files tools/profile_r06.sh left under gpurun_out/ into the committed summaries under profiles/ (run in the
# build container after the GPU calls; the traffic stamps carry the sha256 of the kernel sources of THIS tree, which must be the profiled one).
cd "$(dirname "$0")/.." || exit 1
R=6
python3 tools/make_traffic_json.py r06c2 2 760000000 "tools/pmc_cfg.sh r06c2 64 2 312500 (tools/profile_r06.sh)" $R
python3 tools/make_traffic_json.py r06c3 3 17600000000 "tools/pmc_cfg.sh r06c3 5160 4 77520 (tools/profile_r06.sh)" $R
for spec in "2 38759" "3 58139" "4 77519" "5 96899" "6 116279" "8 155038" "10 193798" "12 232558"; do
  set -- $spec; k=$1; nsh=$2; games=$(( nsh * (5160 / k) ))
  python3 tools/make_traffic_json.py r06s$k 6 $(( games * (34 * k + 40) )) "tools/pmc_cfg.sh r06s$k 5160 $k $nsh" $R $k
done
python3 tools/make_traffic_json.py r06k2 4 27000000000 "tools/pmc_cfg.sh r06k2 5160 2 96899" $R 2
python3 tools/make_traffic_json.py r06k4 4 44000000000 "tools/pmc_cfg.sh r06k4 5160 4 193798" $R 4
python3 tools/make_traffic_json.py r06k6 4 30500000000 "tools/pmc_cfg.sh r06k6 5160 6 145348 (one of the k = 6 call's two launches)" $R 6
python3 tools/make_traffic_json.py r06k8 4 39000000000 "tools/pmc_cfg.sh r06k8 5160 8 193798 (one of the k = 8 call's two launches)" $R 8
python3 tools/make_traffic_json.py r06c5 5 45600000000 "bench.py --config 5 --steps 1 --warmup 0 under tools/pmc_cfg.sh (6 x 10^8 attempts per launch x 76 B)" $R
cp gpurun_out/r06_play_kernel_pmc_summary_config2.txt gpurun_out/r06_play_hc_kernel_pmc_summary_config3.txt gpurun_out/r06_pmc_summary_config6_k*.txt profiles/
for c in 2 3 4 5 6; do cp "$(ls -t gpurun_out/r06_bench_c${c}_stats/*/*_kernel_stats.csv | head -1)" profiles/r06_bench_c${c}_kernel_stats.csv; cp gpurun_out/r06_bench_c${c}_under_rocprof.json profiles/; done
for c in 2 3 4 5 6; do cp gpurun_out/r06_bench_config$c.json profiles/; done
cp gpurun_out/r06_bench.json gpurun_out/r06_farkle_run_end_to_end.json gpurun_out/r06_farkle_run_end_to_end_overlay_disk.json gpurun_out/r06_bench_clock_stamps_off.json gpurun_out/r06_bench_clock_stamps_on.json gpurun_out/r06_bench_driver_command.json gpurun_out/r06_bench_4rank_gloo_one_gpu_config5.json gpurun_out/r06_bench_2rank_gloo_one_gpu.json gpurun_out/r06_bench_2rank_one_gpu_rccl_refused_fallback.json profiles/
ls profiles | grep -c r06
