#!/bin/bash
mkdir -p gpurun_out/r4y
(timeout 1200 python -m pytest tests/test_lstm_gpu.py tests/test_models_gpu.py -x -q -m gpu 2>&1 | tail -3) > gpurun_out/r4y/pytest.txt
for i in 1 2; do
  (TSG_LSTM_SPLITK=0 python bench.py --no-alt --cpu-sample 0 --no-micro --graph on 2>/dev/null | tail -1 | cut -c1-190 | sed "s/^/ONE_GEMM /")
  (python bench.py --no-alt --cpu-sample 0 --no-micro --graph on 2>/dev/null | tail -1 | cut -c1-190 | sed "s/^/SPLIT_K  /")
done > gpurun_out/r4y/bench.txt
python tools/gemm_shapes.py 2>/dev/null | head -8 >> gpurun_out/r4y/bench.txt
cat gpurun_out/r4y/pytest.txt gpurun_out/r4y/bench.txt
