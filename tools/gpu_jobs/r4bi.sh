#!/bin/bash
# GPU job of round 4 (bi): own dropout kernel -- tests + step A/B (TSG_DROPOUT=torch vs own), alternating processes
O=gpurun_out/r4bi; rm -rf $O; mkdir -p $O
(timeout 900 python -m pytest tests/test_dropout_gpu.py tests/test_lstm_gpu.py tests/test_models_gpu.py tests/test_bench_gpu.py -x -q -m gpu 2>&1 | tail -15) > $O/pytest.txt
C="--cpu-sample 0 --no-alt --no-micro --graph on"
for i in 1 2; do
  for d in torch own; do
    echo "== TSG_DROPOUT=$d f32s" >> $O/ab.txt; TSG_DROPOUT=$d python bench.py $C 2>/dev/null | cut -c1-330 >> $O/ab.txt
    echo "== TSG_DROPOUT=$d bf16" >> $O/ab.txt; TSG_DROPOUT=$d python bench.py --dtype bf16 $C 2>/dev/null | cut -c1-330 >> $O/ab.txt
  done
done
cat $O/pytest.txt; grep -o "==.*\|\"value\": [0-9.]*\|\"ms_per_step\": [0-9.]*" $O/ab.txt | paste - - - | head -20
