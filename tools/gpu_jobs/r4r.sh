#!/bin/bash
mkdir -p gpurun_out/r4r
(timeout 1500 python -m pytest tests/test_gemm_f32s_gpu.py tests/test_head_gemm_gpu.py tests/test_wgrad_gpu.py tests/test_lstm_gpu.py tests/test_models_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu 2>&1 | tail -2) > gpurun_out/r4r/pytest.txt
for i in 1 2 3; do
  (python bench.py --no-alt --cpu-sample 0 --no-micro --graph on 2>/dev/null | tail -1 | cut -c1-190)
done > gpurun_out/r4r/bench.txt
(python bench.py --dtype bf16 --no-alt --cpu-sample 0 --no-micro 2>/dev/null | tail -1 | cut -c1-190) >> gpurun_out/r4r/bench.txt
cat gpurun_out/r4r/pytest.txt gpurun_out/r4r/bench.txt
