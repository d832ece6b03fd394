#!/bin/bash
# round 5, session 35: sentence encoder's second LSTM layer backward on the own kernels (dX through the weight-gradient kernel's split scheme): parity + A/B
O=gpurun_out/r5sl; mkdir -p $O
(timeout 2400 python -m pytest tests/test_fullsize_gpu.py tests/test_models_gpu.py tests/test_lstm_gpu.py tests/test_config4_gpu.py -q -m gpu 2>&1 | grep "passed\|failed\|^E " | head -8) > $O/pytest.txt; cat $O/pytest.txt
for rep in 1 2 3; do
  for v in 1 0; do
    echo "own=$v f32s: $(TSG_LSTM_SPLITK=$v python bench.py --no-alt --cpu-sample 0 --no-micro --graph on 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])')" >> $O/bench.txt
  done
done
sort $O/bench.txt
