#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
echo "== pytest -m gpu" && timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r5f_pytest.log 2>&1; rc=$?; tail -3 gpurun_out/r5f_pytest.log
[ $rc -ne 0 ] && exit $rc
for c in 2 6 4; do
  echo "== bench config $c"; timeout -k 10 600 python bench.py --config $c > gpurun_out/r5f_bench_c$c.json 2> gpurun_out/r5f_bench_c$c.err; echo "rc $?"
  python - <<PY
import json
l=json.loads(open('gpurun_out/r5f_bench_c$c.json').read().strip().splitlines()[-1]); r=l['roofline']
print('value %.4g ms/step %.2f frac %.3f kernel %s kernel_ms %.2f clock %s frac@clock %s' % (l['value'], l['ms_per_step'], r['frac'], r['kernel'], r['kernel_ms'], r.get('clock_mhz_measured'), r.get('frac_at_measured_clock')))
for p in r.get('per_k', []):
    print('   k=%d %s block %s ms %.1f games/s %.4g W %.0f frac %.3f clock %s' % (p['k'], p['kernel'], p['launch']['play_block'], p['kernel_ms'], p['kernel_games_per_s'], p['ops_per_game'], p['frac'], p.get('clock_mhz_measured')))
print('cpu', l.get('cpu_baseline', {}) and l['cpu_baseline'].get('value'))
PY
done
