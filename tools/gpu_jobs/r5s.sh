#!/bin/bash
# round 5, session 19: time-major vs batch-major sequence tensors under the shipped persistent kernels (is the layout what the streams wait for?)
O=gpurun_out/r5s; mkdir -p $O
export TSG_STEPK=0
for rep in 1 2; do
for dt in 2 1; do
  for SHAPE in "128 128 512" "64 128 512"; do
    for bm in 1 0; do
      echo "== dt=$dt [$SHAPE] batch-major=$bm" >> $O/lstm_ab.txt
      TSG_REC_DTYPE=$dt TSG_BM=$bm python -u tools/lstm_bench.py $SHAPE 2>&1 | grep -v amdgpu | grep "persistent backward\|rec dtype" | sed 's/, err word.*//' >> $O/lstm_ab.txt
    done
  done
done
done
cat $O/lstm_ab.txt
