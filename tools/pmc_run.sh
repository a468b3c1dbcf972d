#!/bin/bash
# usage: tools/pmc_run.sh <outdir> <counter list...>   (one PMC pass; kernel-trace only)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=$1; shift
rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$out" -- python3 tools/time_calls.py 312500 > "$out.log" 2>&1
