#!/bin/bash
mkdir -p gpurun_out/r4an
(timeout 2400 python -m pytest tests/test_wgrad_gpu.py tests/test_bf16_storage_gpu.py tests/test_config5_bf16_gpu.py tests/test_lstm_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu 2>&1 | tail -3) > gpurun_out/r4an/pytest.txt
for i in 1 2; do
(TSG_WGRAD_BF16_TR=0 python bench.py --dtype bf16 --graph on --no-alt --cpu-sample 0 --no-micro 2>/dev/null | tail -1 | cut -c1-200 | sed "s/^/REG_STAGED /")
(python bench.py --dtype bf16 --graph on --no-alt --cpu-sample 0 --no-micro 2>/dev/null | tail -1 | cut -c1-200 | sed "s/^/DMA_TR     /")
done > gpurun_out/r4an/bench.txt
cat gpurun_out/r4an/pytest.txt gpurun_out/r4an/bench.txt
