#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2_perm_stats -- python3 tools/time_config.py 5160 4 77520 2 0 0 > gpurun_out/r2_perm_stats.log 2>&1
python3 - <<'PY'
import csv, glob
for f in glob.glob('gpurun_out/r2_perm_stats/**/*_kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        print(f"{r['Name'][:60]:60s} calls {r['Calls']:>3s} avg {float(r['AverageNs'])/1e6:8.3f} ms")
PY
