#!/bin/bash
# K1g forward at the other configs' shapes, the tree's library vs variants.   usage: k1_shapes_ab.sh OUT name,name
O=gpurun_out/$1; mkdir -p $O
for rep in 1 2; do
for shape in "64 256 25" "128 512 25" "32 64 20" "64 128 15"; do
  echo "== tree [$shape]" >> $O/k1.txt; python tools/k1_fwd_modes_time.py $shape 2>/dev/null | grep "f32s:\|bf16:" >> $O/k1.txt
  for v in ${2//,/ }; do
    echo "== $v [$shape]" >> $O/k1.txt; TSG_HIP_LIB=tools/_ablate/$v.so python tools/k1_fwd_modes_time.py $shape 2>/dev/null | grep "f32s:\|bf16:" >> $O/k1.txt
  done
done
done
paste - - - < $O/k1.txt | cut -c1-200
