#!/bin/bash
# round 5, first GPU call: parity suite on the hand-laid generator step, bench line, A/B against the compiler's lowering, issue rates
set -o pipefail
mkdir -p gpurun_out
echo "== pytest -m gpu" && timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r5a_pytest.log 2>&1; rc=$?; tail -3 gpurun_out/r5a_pytest.log
[ $rc -ne 0 ] && exit $rc
echo "== bench config 2" && timeout -k 10 300 python bench.py --steps 20 --warmup 2 > gpurun_out/r5a_bench_c2.json 2> gpurun_out/r5a_bench_c2.err && python - <<'PY'
import json; l=json.loads(open('gpurun_out/r5a_bench_c2.json').read().strip().splitlines()[-1]); r=l['roofline']
print('value', l['value'], 'ms/step', l['ms_per_step'], 'frac', r['frac'], 'kernel_ms', r['kernel_ms'], 'clock', r.get('clock_mhz_measured'), 'frac@clock', r.get('frac_at_measured_clock'))
PY
echo "== A/B k=2 64-grid" && bash tools/ab_run.sh "python tools/time_config.py 64 2 312500 4" plain asm 2 > gpurun_out/r5a_ab_k2.log 2>&1; cat gpurun_out/r5a_ab_k2.log
echo "== A/B k=4 5160" && bash tools/ab_run.sh "python tools/time_config.py 5160 4 4000 3" plain asm 1 > gpurun_out/r5a_ab_k4.log 2>&1; cat gpurun_out/r5a_ab_k4.log
echo "== A/B k=8 5160" && bash tools/ab_run.sh "python tools/time_config.py 5160 8 4000 3" plain asm 1 > gpurun_out/r5a_ab_k8.log 2>&1; cat gpurun_out/r5a_ab_k8.log
echo "== issue rates" && timeout -k 10 200 tools/valu_rates 8 150000 > gpurun_out/r5a_valu_rates_8w.txt 2>&1; cat gpurun_out/r5a_valu_rates_8w.txt
timeout -k 10 200 tools/valu_rates 4 150000 > gpurun_out/r5a_valu_rates_4w.txt 2>&1; tail -30 gpurun_out/r5a_valu_rates_4w.txt
