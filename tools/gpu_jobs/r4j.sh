#!/bin/bash
# GPU job of round 4 (j): contraction-major GEMM operand (tsg_gemm_f32s_nn: no transposed weight copies) + two-output weight gradient
# (tsg_wgrad_f32s_out2: no slicing copies): parity, then the step time A/B against TSG_NO_COPIES=0.
mkdir -p gpurun_out/r4j
(timeout 900 python -m pytest tests/test_gemm_f32s_gpu.py tests/test_wgrad_gpu.py tests/test_head_gemm_gpu.py tests/test_lstm_gpu.py tests/test_models_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu 2>&1 | tail -6) > gpurun_out/r4j/pytest.txt
for i in 1 2 3; do
  (TSG_NO_COPIES=0 python bench.py --no-alt --cpu-sample 0 --no-micro --graph on 2>/dev/null | tail -1 | cut -c1-200 | sed "s/^/COPIES    /")
  (python bench.py --no-alt --cpu-sample 0 --no-micro --graph on 2>/dev/null | tail -1 | cut -c1-200 | sed "s/^/NOCOPIES  /")
done > gpurun_out/r4j/bench_no_copies_ab.txt
cat gpurun_out/r4j/pytest.txt gpurun_out/r4j/bench_no_copies_ab.txt
