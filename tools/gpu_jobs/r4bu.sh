#!/bin/bash
# 2-slot ring as the default: LSTM tests, soak, whole suite, step A/B vs TSG_LSTM_RING=4
O=gpurun_out/r4bu; rm -rf $O; mkdir -p $O
(timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -4) > $O/pytest_gpu_full.txt
timeout 1500 python tools/lstm_soak.py 300 > $O/lstm_soak.txt 2>&1
C="--cpu-sample 0 --no-alt --no-micro --graph on"
for i in 1 2; do for r in 4 2; do
  echo "== TSG_LSTM_RING=$r f32s" >> $O/ab.txt; TSG_LSTM_RING=$r python bench.py $C 2>/dev/null | cut -c1-330 >> $O/ab.txt
  echo "== TSG_LSTM_RING=$r bf16" >> $O/ab.txt; TSG_LSTM_RING=$r python bench.py --dtype bf16 $C 2>/dev/null | cut -c1-330 >> $O/ab.txt
done; done
for r in 4 2; do
  echo "== TSG_LSTM_RING=$r config 3 shape f32s" >> $O/ab.txt; TSG_LSTM_RING=$r python bench.py --B 64 --T 256 --N 25 $C 2>/dev/null | cut -c1-330 >> $O/ab.txt
  echo "== TSG_LSTM_RING=$r config 4 shard bf16" >> $O/ab.txt; TSG_LSTM_RING=$r python bench.py --B 16 --T 512 --N 25 --dtype bf16 $C 2>/dev/null | cut -c1-330 >> $O/ab.txt
done
cat $O/pytest_gpu_full.txt; grep -v amdgpu $O/lstm_soak.txt | tail -3; grep -o "==.*\|\"ms_per_step\": [0-9.]*" $O/ab.txt | paste - -
