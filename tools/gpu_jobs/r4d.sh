#!/bin/bash
# GPU job of round 4 (d): MFMA-shape A/B of the own GEMM, parity of the fused nodes, A/B of the fused recalibration node in the step,
# then the round-4 profiling session (tools/profile_r4.sh).
mkdir -p gpurun_out/r4d
(LIB=1 python tools/gemm_mfma_shape_ab.py; TSG_GEMM_MFMA16=1 python tools/gemm_mfma_shape_ab.py; python tools/gemm_mfma_shape_ab.py; TSG_GEMM_MFMA16=1 python tools/gemm_mfma_shape_ab.py) > gpurun_out/r4d/gemm_mfma_shape_ab.txt 2>&1
(timeout 900 python -m pytest tests/test_head_gemm_gpu.py tests/test_gemm_f32s_gpu.py tests/test_models_gpu.py tests/test_fullsize_gpu.py tests/test_config1_gpu.py -x -q -m gpu 2>&1 | tail -25) > gpurun_out/r4d/pytest_subset.txt
for i in 1 2; do
  (python bench.py --no-alt --cpu-sample 0 --no-micro --graph on 2>/dev/null | tail -1 | cut -c1-330 | sed "s/^/FUSED  /")
  (TSG_RECAL_FUSED=0 python bench.py --no-alt --cpu-sample 0 --no-micro --graph on 2>/dev/null | tail -1 | cut -c1-330 | sed "s/^/UNFUSED /")
done > gpurun_out/r4d/bench_recal_ab.txt
bash tools/profile_r4.sh > gpurun_out/r4d/profile_r4.log 2>&1
cat gpurun_out/r4d/gemm_mfma_shape_ab.txt; cat gpurun_out/r4d/pytest_subset.txt; cat gpurun_out/r4d/bench_recal_ab.txt; tail -60 gpurun_out/r4d/profile_r4.log
