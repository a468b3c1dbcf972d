#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
{ for lib in fused fusedlt; do echo "== $lib k=6"; FARKLE_HIP_LIB=$PWD/tools/ab/lib_$lib.so timeout -k 10 200 python tools/time_config.py 5160 6 4000 4 clock_stamps=1; done
  for k in 5 8 10 12; do echo "== fusedlt k=$k"; FARKLE_HIP_LIB=$PWD/tools/ab/lib_fusedlt.so timeout -k 10 200 python tools/time_config.py 5160 $k 4000 3 clock_stamps=1; done
  echo "== fusedlt k=2/3/4"; for k in 2 3 4; do FARKLE_HIP_LIB=$PWD/tools/ab/lib_fusedlt.so timeout -k 10 200 python tools/time_config.py 5160 $k 4000 3 clock_stamps=1; done
  echo "== fusedlt k=2 64 grid 10^7"; FARKLE_HIP_LIB=$PWD/tools/ab/lib_fusedlt.so timeout -k 10 200 python tools/time_config.py 64 2 312500 3 clock_stamps=1
  echo "== rows of rep 2 of k=6 (shuffles 8000..12000)"; FARKLE_HIP_LIB=$PWD/tools/ab/lib_fusedlt.so timeout -k 10 200 python tools/time_config.py 5160 6 4000 3 0 1
} > gpurun_out/r5j_k6.log 2>&1
cut -c1-30,60-250 gpurun_out/r5j_k6.log
