#!/bin/bash
mkdir -p gpurun_out/r4x
python tools/gemm_small_time.py 2>&1 | grep -v amdgpu > gpurun_out/r4x/small.txt
(timeout 2400 python -m pytest tests/test_gemm_f32s_gpu.py tests/test_head_gemm_gpu.py tests/test_scdm_gpu.py tests/test_models_gpu.py tests/test_fullsize_gpu.py tests/test_config1_gpu.py tests/test_lstm_gpu.py -x -q -m gpu 2>&1 | tail -4) > gpurun_out/r4x/pytest.txt
for i in 1 2 3; do
  (python bench.py --no-alt --cpu-sample 0 --no-micro --graph on 2>/dev/null | tail -1 | cut -c1-190)
done > gpurun_out/r4x/bench.txt
python tools/gemm_shapes.py 2>/dev/null | head -12 >> gpurun_out/r4x/bench.txt
cat gpurun_out/r4x/small.txt gpurun_out/r4x/pytest.txt gpurun_out/r4x/bench.txt
