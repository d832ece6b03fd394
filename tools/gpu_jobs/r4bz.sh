#!/bin/bash
# product build (nt: backward streaming operands; forward R / Cs stores in the bf16 mode) vs -DTSG_LSTM_NO_NT: tests + bf16 / f32s step
O=$PWD/gpurun_out/r4bz; rm -rf $O; mkdir -p $O
NONT=$PWD/tools/_ablate/lstm_nont.so
(timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -3) > $O/pytest_gpu_full.txt
C="--cpu-sample 0 --no-alt --no-micro --graph on"
for i in 1 2 3; do for v in nont nt; do
  if [ $v = nont ]; then export TSG_HIP_LIB=$NONT; else unset TSG_HIP_LIB; fi
  echo "== $v bf16" >> $O/ab.txt; python bench.py --dtype bf16 $C 2>/dev/null | cut -c1-330 >> $O/ab.txt
  echo "== $v bf16 config 4 shard" >> $O/ab.txt; python bench.py --dtype bf16 --B 16 --T 512 --N 25 $C 2>/dev/null | cut -c1-330 >> $O/ab.txt
  echo "== $v bf16 config 3 shape" >> $O/ab.txt; python bench.py --dtype bf16 --B 64 --T 256 --N 25 $C 2>/dev/null | cut -c1-330 >> $O/ab.txt
done; done
unset TSG_HIP_LIB
cat $O/pytest_gpu_full.txt; grep -o "==.*\|\"ms_per_step\": [0-9.]*" $O/ab.txt | paste - -
