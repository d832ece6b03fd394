#!/bin/bash
# K1g forward diagnosis: ticks of priority-swapped / store-less builds + plain timing of the store-less build
O=gpurun_out/$1; mkdir -p $O
for v in k1t_prio k1t_nost; do
  echo "=== $v gate=1" >> $O/ticks.txt
  TSG_HIP_LIB=tools/_ablate/$v.so python tools/k1_ticks.py 128 1 2 >> $O/ticks.txt 2>&1
done
for v in k1_nost; do
  echo "=== $v" >> $O/ticks.txt
  TSG_HIP_LIB=tools/_ablate/$v.so python tools/k1_variants.py 128 200 2>&1 | grep "^f32s\|^bf16" >> $O/ticks.txt
done
echo "=== tree" >> $O/ticks.txt
python tools/k1_variants.py 128 200 2>&1 | grep "^f32s\|^bf16" >> $O/ticks.txt
cat $O/ticks.txt
