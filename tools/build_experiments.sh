#!/bin/bash
# The library WITH the kernel instances and fk_set_option names that lost or tied against the launch plan's choices (profiles/HISTORY.md): tools/ab/lib_experiments.so, selected with FARKLE_HIP_LIB=$PWD/tools/ab/lib_experiments.so.  The
# tools/exp_*.py scripts that produced the logs need it; the shipped farkle_ii_amd/libfarkle_hip.so does not contain them.
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DFK_EXPERIMENTS -o tools/ab/lib_experiments.so farkle_ii_amd/csrc/farkle_hip.hip
ls -la tools/ab/lib_experiments.so
