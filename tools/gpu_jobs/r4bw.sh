#!/bin/bash
# LSTM forward + backward with the non-temporal hint on the streaming operands (-DTSG_LSTM_NT build) vs the product build: tests, per-step times, train step
O=$PWD/gpurun_out/r4bw; rm -rf $O; mkdir -p $O
NT=$PWD/tools/_ablate/lstm_nt.so
(TSG_HIP_LIB=$NT timeout 900 python -m pytest tests/test_lstm_gpu.py tests/test_bf16_storage_gpu.py -x -q -m gpu 2>&1 | tail -2) > $O/pytest_nt.txt
for shape in "128 128 512" "128 256 512" "64 20 512" "32 512 512"; do for dt in 2 1; do for v in base nt base nt; do
  echo "== $shape dtype $dt $v" >> $O/nt.txt
  if [ $v = nt ]; then export TSG_HIP_LIB=$NT; else unset TSG_HIP_LIB; fi
  TSG_REC_DTYPE=$dt TSG_BM=1 timeout 300 python tools/lstm_bench.py $shape 2>&1 | grep "persistent backward\|rec dtype" | cut -c1-60 | paste - - | cut -c37-60,97-130 >> $O/nt.txt
done; done; done
unset TSG_HIP_LIB
C="--cpu-sample 0 --no-alt --no-micro --graph on"
for i in 1 2; do for v in base nt; do
  if [ $v = nt ]; then export TSG_HIP_LIB=$NT; else unset TSG_HIP_LIB; fi
  echo "== $v f32s" >> $O/ab.txt; python bench.py $C 2>/dev/null | cut -c1-330 >> $O/ab.txt
  echo "== $v bf16" >> $O/ab.txt; python bench.py --dtype bf16 $C 2>/dev/null | cut -c1-330 >> $O/ab.txt
done; done
cat $O/pytest_nt.txt; paste - - < $O/nt.txt | cut -c1-130; grep -o "==.*\|\"ms_per_step\": [0-9.]*" $O/ab.txt | paste - -
