#!/bin/bash
# fixed cost of a persistent LSTM launch: time vs T at [128, T, 512], f32s and bf16, batch-major
O=gpurun_out/r4bm; rm -rf $O; mkdir -p $O
for dt in 2 1; do for T in 8 16 32 64 128 256; do
  TSG_REC_DTYPE=$dt TSG_BM=1 timeout 300 python tools/lstm_bench.py 128 $T 512 2>&1 | grep "rec dtype\|persistent backward" | cut -c1-110 >> $O/lstm_vs_T.txt
done; done
cat $O/lstm_vs_T.txt
