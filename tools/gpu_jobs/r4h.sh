#!/bin/bash
# GPU job of round 4 (h): the 256 x 256 tile variant of the weight-gradient kernel (default where N, K0, K1 are multiples of 256) vs the
# 256 x 128 kernel (TSG_WGRAD_TILE=128): parity, stand-alone timing, step time A/B.
mkdir -p gpurun_out/r4h
(timeout 900 python -m pytest tests/test_wgrad_gpu.py tests/test_lstm_gpu.py tests/test_models_gpu.py tests/test_head_gemm_gpu.py -x -q -m gpu 2>&1 | tail -6) > gpurun_out/r4h/pytest_wgrad256.txt
for i in 1 2; do
  (python tools/wgrad_time.py 2>&1 | cut -c1-150 | sed "s/^/T256  /")
  (TSG_WGRAD_TILE=128 python tools/wgrad_time.py 2>&1 | cut -c1-150 | sed "s/^/T128  /")
done > gpurun_out/r4h/wgrad_tile_standalone.txt
for i in 1 2 3; do
  (TSG_WGRAD_TILE=128 python bench.py --no-alt --cpu-sample 0 --no-micro --graph on 2>/dev/null | tail -1 | cut -c1-200 | sed "s/^/T128  /")
  (python bench.py --no-alt --cpu-sample 0 --no-micro --graph on 2>/dev/null | tail -1 | cut -c1-200 | sed "s/^/T256  /")
done > gpurun_out/r4h/bench_wgrad_tile_ab.txt
cat gpurun_out/r4h/pytest_wgrad256.txt gpurun_out/r4h/wgrad_tile_standalone.txt gpurun_out/r4h/bench_wgrad_tile_ab.txt
