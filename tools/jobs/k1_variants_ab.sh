#!/bin/bash
# K1g forward stand-alone timing of the tree's library vs variant libraries (tools/_ablate/<name>.so), alternating processes on one box.
#   usage: k1_variants_ab.sh OUT name,name,... [reps]
O=gpurun_out/$1; mkdir -p $O
for rep in $(seq 1 ${3:-3}); do
  echo "== tree" >> $O/k1.txt; python tools/k1_variants.py 128 200 2>&1 | grep "^f32s\|^bf16" >> $O/k1.txt
  for v in ${2//,/ }; do
    echo "== $v" >> $O/k1.txt; TSG_HIP_LIB=tools/_ablate/$v.so python tools/k1_variants.py 128 200 2>&1 | grep "^f32s\|^bf16" >> $O/k1.txt
  done
done
grep "==\|^f32s  \|^bf16  " $O/k1.txt | paste - - - - | cut -c1-250
