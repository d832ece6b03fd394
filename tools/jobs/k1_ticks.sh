#!/bin/bash
# K1g forward: per-phase tick sums of an instrumented build.   usage: k1_ticks.sh OUTDIR lib...
O=gpurun_out/$1; shift; mkdir -p $O
for v in "$@"; do
  for g in 1 0; do
    echo "=== $v gate=$g" >> $O/ticks.txt
    TSG_HIP_LIB=tools/_ablate/$v.so python tools/k1_ticks.py 128 $g 2 >> $O/ticks.txt 2>&1
  done
done
cat $O/ticks.txt
