#!/bin/bash
# GPU job of round 4 (bc): padded persistent-LSTM grids (whole exchange groups on one XCD) vs natural grids at small batches
O=gpurun_out/r4bc; rm -rf $O; mkdir -p $O
(timeout 900 python -m pytest tests/test_lstm_gpu.py -x -q -m gpu 2>&1 | tail -3) > $O/pytest_lstm.txt
for shape in "32 512 512" "40 128 512" "96 128 512" "64 256 512" "128 128 512" "32 128 256"; do
  for dt in 2 1; do
    for pad in 0 1 0 1; do
      echo "== B T h = $shape dtype $dt TSG_LSTM_PAD=$pad" >> $O/lstm_pad_ab.txt
      TSG_LSTM_PAD=$pad TSG_REC_DTYPE=$dt TSG_BM=1 timeout 300 python tools/lstm_bench.py $shape 2>&1 | grep -v "^sync word\|amdgpu.ids" | cut -c1-330 >> $O/lstm_pad_ab.txt
    done
  done
done
cat $O/pytest_lstm.txt; grep "==\|rec dtype\|persistent backward" $O/lstm_pad_ab.txt | cut -c1-150
