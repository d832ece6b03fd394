#!/bin/bash
# GPU job of round 4 (g): own GEMM with staggered SIMD partners (TSG_GEMM_STAGGER=1: waves 4..7 multiply first, convert after) vs the
# lock-step order: parity of the GEMM tests with the switch on, the step time A/B, stand-alone GEMM timing.
mkdir -p gpurun_out/r4g
(TSG_GEMM_STAGGER=1 timeout 600 python -m pytest tests/test_gemm_f32s_gpu.py tests/test_head_gemm_gpu.py tests/test_lstm_gpu.py -x -q -m gpu 2>&1 | tail -4) > gpurun_out/r4g/pytest_stagger.txt
for i in 1 2 3; do
  (TSG_GEMM_STAGGER=0 python bench.py --no-alt --cpu-sample 0 --no-micro --graph on 2>/dev/null | tail -1 | cut -c1-200 | sed "s/^/LOCKSTEP  /")
  (TSG_GEMM_STAGGER=1 python bench.py --no-alt --cpu-sample 0 --no-micro --graph on 2>/dev/null | tail -1 | cut -c1-200 | sed "s/^/STAGGER   /")
done > gpurun_out/r4g/bench_gemm_stagger_ab.txt
for i in 1 2; do
  (TSG_GEMM_STAGGER=0 TSG_GEMM=1 python tools/gemm_f32s_time.py 2>&1 | grep "own" | cut -c1-110 | sed "s/^/LOCKSTEP  /")
  (TSG_GEMM_STAGGER=1 TSG_GEMM=1 python tools/gemm_f32s_time.py 2>&1 | grep "own" | cut -c1-110 | sed "s/^/STAGGER   /")
done > gpurun_out/r4g/gemm_stagger_standalone.txt
cat gpurun_out/r4g/pytest_stagger.txt gpurun_out/r4g/bench_gemm_stagger_ab.txt gpurun_out/r4g/gemm_stagger_standalone.txt
