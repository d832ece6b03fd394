#!/bin/bash
mkdir -p gpurun_out/r4o
(timeout 900 python -m pytest tests/test_wgrad_gpu.py tests/test_lstm_gpu.py -x -q -m gpu 2>&1 | tail -2) > gpurun_out/r4o/pytest.txt
python tools/wgrad_lstm_probe.py > gpurun_out/r4o/probe.txt 2>&1
for i in 1 2 3; do
  (python bench.py --no-alt --cpu-sample 0 --no-micro --graph on 2>/dev/null | tail -1 | cut -c1-190)
done > gpurun_out/r4o/bench.txt
cat gpurun_out/r4o/pytest.txt gpurun_out/r4o/probe.txt gpurun_out/r4o/bench.txt
