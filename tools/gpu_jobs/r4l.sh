#!/bin/bash
# GPU job of round 4 (l): the LayerNorm kernels: parity, then the step time A/B (TSG_LN=0 keeps torch's LayerNorm), both modes.
mkdir -p gpurun_out/r4l
(timeout 900 python -m pytest tests/test_layer_norm_gpu.py tests/test_models_gpu.py tests/test_fullsize_gpu.py tests/test_config1_gpu.py -x -q -m gpu 2>&1 | tail -6) > gpurun_out/r4l/pytest.txt
for i in 1 2 3; do
  (TSG_LN=0 python bench.py --no-alt --cpu-sample 0 --no-micro --graph on 2>/dev/null | tail -1 | cut -c1-190 | sed "s/^/TORCH_LN  /")
  (python bench.py --no-alt --cpu-sample 0 --no-micro --graph on 2>/dev/null | tail -1 | cut -c1-190 | sed "s/^/OWN_LN    /")
done > gpurun_out/r4l/bench_ln_ab.txt
(TSG_LN=0 python bench.py --dtype bf16 --no-alt --cpu-sample 0 --no-micro 2>/dev/null | tail -1 | cut -c1-190 | sed "s/^/TORCH_LN bf16  /"; python bench.py --dtype bf16 --no-alt --cpu-sample 0 --no-micro 2>/dev/null | tail -1 | cut -c1-190 | sed "s/^/OWN_LN bf16    /") >> gpurun_out/r4l/bench_ln_ab.txt
cat gpurun_out/r4l/pytest.txt gpurun_out/r4l/bench_ln_ab.txt
