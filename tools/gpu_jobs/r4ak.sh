#!/bin/bash
mkdir -p gpurun_out/r4ak
(timeout 1800 python -m pytest tests/test_bf16_storage_gpu.py tests/test_config5_bf16_gpu.py tests/test_wgrad_gpu.py tests/test_lstm_gpu.py -x -q -m gpu 2>&1 | tail -3) > gpurun_out/r4ak/pytest.txt
for i in 1 2 3; do
(python bench.py --dtype bf16 --graph on --no-alt --cpu-sample 0 --no-micro 2>/dev/null | tail -1 | cut -c1-420)
done > gpurun_out/r4ak/bench.txt
cat gpurun_out/r4ak/pytest.txt gpurun_out/r4ak/bench.txt
