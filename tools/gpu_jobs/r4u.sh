#!/bin/bash
mkdir -p gpurun_out/r4u
(timeout 2400 python -m pytest tests/test_lstm_gpu.py tests/test_models_gpu.py tests/test_fullsize_gpu.py tests/test_config1_gpu.py tests/test_scdm_gpu.py tests/test_bf16_storage_gpu.py tests/test_bench_gpu.py tests/test_config4_gpu.py -x -q -m gpu 2>&1 | tail -8) > gpurun_out/r4u/pytest.txt
for i in 1 2 3; do
  (python bench.py --no-alt --cpu-sample 0 --no-micro --graph on 2>/dev/null | tail -1 | cut -c1-190)
done > gpurun_out/r4u/bench.txt
(python bench.py --dtype bf16 --no-alt --cpu-sample 0 --no-micro 2>/dev/null | tail -1 | cut -c1-190) >> gpurun_out/r4u/bench.txt
python tools/glue_sites.py 2>/dev/null | head -2 >> gpurun_out/r4u/bench.txt
cat gpurun_out/r4u/pytest.txt gpurun_out/r4u/bench.txt
