#!/bin/bash
mkdir -p gpurun_out/r4aa
(timeout 2400 python -m pytest tests/test_models_gpu.py tests/test_fullsize_gpu.py tests/test_bench_gpu.py tests/test_input_pipeline_gpu.py tests/test_config4_gpu.py -x -q -m gpu 2>&1 | tail -3) > gpurun_out/r4aa/pytest.txt
for i in 1 2 3; do
  (python bench.py --no-alt --cpu-sample 0 --no-micro --graph on 2>/dev/null | tail -1 | cut -c1-190)
done > gpurun_out/r4aa/bench.txt
python tools/glue_sites.py 2>/dev/null | head -4 >> gpurun_out/r4aa/bench.txt
cat gpurun_out/r4aa/pytest.txt gpurun_out/r4aa/bench.txt
