#!/bin/bash
# GPU job of round 4 (t): glue reductions (own transpose, final-state gather, three-output MomentPooling, parameter-level heads)
mkdir -p gpurun_out/r4t
(timeout 2400 python -m pytest tests/test_transpose_gpu.py tests/test_head_gemm_gpu.py tests/test_moment_pool_gpu.py tests/test_lstm_gpu.py tests/test_models_gpu.py tests/test_fullsize_gpu.py tests/test_config1_gpu.py tests/test_match_head_gpu.py tests/test_boundary_gpu.py tests/test_bf16_storage_gpu.py -x -q -m gpu 2>&1 | tail -12) > gpurun_out/r4t/pytest.txt
for i in 1 2; do
  (TSG_TRANSPOSE=torch python bench.py --no-alt --cpu-sample 0 --no-micro --graph on 2>/dev/null | tail -1 | cut -c1-190 | sed "s/^/TORCH_T /")
  (python bench.py --no-alt --cpu-sample 0 --no-micro --graph on 2>/dev/null | tail -1 | cut -c1-190 | sed "s/^/OWN_T   /")
done > gpurun_out/r4t/bench.txt
(python bench.py --dtype bf16 --no-alt --cpu-sample 0 --no-micro 2>/dev/null | tail -1 | cut -c1-190) >> gpurun_out/r4t/bench.txt
python tools/glue_sites.py 2>/dev/null | head -3 >> gpurun_out/r4t/bench.txt
cat gpurun_out/r4t/pytest.txt gpurun_out/r4t/bench.txt
