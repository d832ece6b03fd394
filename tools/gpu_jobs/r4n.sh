#!/bin/bash
# GPU job of round 4 (n): branchless shifted segment of the weight-gradient kernel: parity, the shape probe again, the step
mkdir -p gpurun_out/r4n
(timeout 900 python -m pytest tests/test_wgrad_gpu.py tests/test_lstm_gpu.py tests/test_models_gpu.py -x -q -m gpu 2>&1 | tail -4) > gpurun_out/r4n/pytest.txt
python tools/wgrad_lstm_probe.py > gpurun_out/r4n/probe.txt 2>&1
for i in 1 2 3; do
  (python bench.py --no-alt --cpu-sample 0 --no-micro --graph on 2>/dev/null | tail -1 | cut -c1-190)
done > gpurun_out/r4n/bench.txt
(python bench.py --dtype bf16 --no-alt --cpu-sample 0 --no-micro 2>/dev/null | tail -1 | cut -c1-190) >> gpurun_out/r4n/bench.txt
cat gpurun_out/r4n/pytest.txt gpurun_out/r4n/probe.txt gpurun_out/r4n/bench.txt
