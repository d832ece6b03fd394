#!/bin/bash
mkdir -p gpurun_out/r4z
python tools/gemm_f32s_time.py 2>&1 | grep -v amdgpu > gpurun_out/r4z/gemm.txt
(timeout 1200 python -m pytest tests/test_lstm_gpu.py tests/test_match_head_gpu.py tests/test_models_gpu.py -x -q -m gpu 2>&1 | tail -2) > gpurun_out/r4z/pytest.txt
for i in 1 2 3; do
  (python bench.py --no-alt --cpu-sample 0 --no-micro --graph on 2>/dev/null | tail -1 | cut -c1-190)
done > gpurun_out/r4z/bench.txt
cat gpurun_out/r4z/gemm.txt gpurun_out/r4z/pytest.txt gpurun_out/r4z/bench.txt
