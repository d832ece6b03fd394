#!/bin/bash
# GPU job of round 4 (bd): final padded-grid policy (tests) + 16-unit forward workgroups (TSG_LSTM_NW=4) at small batches
O=gpurun_out/r4bd; rm -rf $O; mkdir -p $O
(timeout 900 python -m pytest tests/test_lstm_gpu.py -x -q -m gpu 2>&1 | tail -3) > $O/pytest_lstm.txt
(TSG_LSTM_NW=4 timeout 900 python -m pytest tests/test_lstm_gpu.py -x -q -m gpu 2>&1 | tail -3) > $O/pytest_lstm_nw4.txt
for shape in "32 512 512" "40 128 512" "16 512 512" "48 256 512"; do
  for nw in 8 4 8 4; do
    echo "== B T h = $shape f32s TSG_LSTM_NW=$nw" >> $O/lstm_nw_ab.txt
    TSG_LSTM_NW=$nw TSG_REC_DTYPE=2 TSG_BM=1 timeout 300 python tools/lstm_bench.py $shape 2>&1 | grep -v "amdgpu.ids" | cut -c1-200 >> $O/lstm_nw_ab.txt
  done
done
cat $O/pytest_lstm.txt $O/pytest_lstm_nw4.txt; grep "==\|rec dtype\|^sync" $O/lstm_nw_ab.txt | cut -c1-150
