#!/bin/bash
# A/B of two builds of the library in alternating processes (same box, same call): tools/ab/lib_<name>.so, selected through FARKLE_HIP_LIB.
# usage: tools/ab_run.sh "<python command>" name1 name2 [rounds=2]
cmd=$1; a=$2; b=$3; n=${4:-2}
for i in $(seq 1 $n); do
  for v in $a $b; do
    echo "== $v (round $i)"
    FARKLE_HIP_LIB=$PWD/tools/ab/lib_$v.so timeout -k 10 300 $cmd 2>&1 | grep -v "^R=" | tail -2
  done
done
