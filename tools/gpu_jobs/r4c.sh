#!/bin/bash
# GPU job of round 4 (c): parity subset after the fused heads / MomentPooling / packed-f16 K1g, K1g backward A/B (with / without the
# acknowledgement wait), K1g forward in the three dtypes, default and bf16 bench lines.
mkdir -p gpurun_out/r4c
(timeout 1200 python -m pytest tests/test_head_gemm_gpu.py tests/test_moment_pool_gpu.py tests/test_scdm_gpu.py tests/test_bf16_storage_gpu.py tests/test_models_gpu.py tests/test_fullsize_gpu.py tests/test_config5_bf16_gpu.py -x -q -m gpu 2>&1 | tail -40) > gpurun_out/r4c/pytest_subset.txt
for i in 1 2; do
  python tools/k1_bwd_time.py 128 2>&1 | tail -1
  TSG_HIP_LIB=$PWD/tools/_ablate/k1_noack.so python tools/k1_bwd_time.py 128 2>&1 | tail -1 | sed "s/^/NOACK /"
done > gpurun_out/r4c/k1_bwd_ab.txt
(python tools/k1_fwd_modes_time.py 128 128 20; python tools/k1_fwd_modes_time.py 64 256 25) > gpurun_out/r4c/k1_fwd_modes.txt 2>&1
(python bench.py 2>gpurun_out/r4c/bench_err.txt | tail -1) > gpurun_out/r4c/bench_default.json
(python bench.py --dtype bf16 --no-alt 2>gpurun_out/r4c/bench_bf16_err.txt | tail -1) > gpurun_out/r4c/bench_bf16.json
tail -30 gpurun_out/r4c/pytest_subset.txt; cat gpurun_out/r4c/k1_bwd_ab.txt gpurun_out/r4c/k1_fwd_modes.txt
cut -c1-300 gpurun_out/r4c/bench_default.json; echo; cut -c1-300 gpurun_out/r4c/bench_bf16.json
