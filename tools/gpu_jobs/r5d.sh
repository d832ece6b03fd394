#!/bin/bash
# round 5, session 4: own bf16 GEMM (parity, timing, bf16 step A/B), stream-K tests after the fix, LSTM ring policy, config-5 norms
O=gpurun_out/r5d; mkdir -p $O
(timeout 1500 python -m pytest tests/test_gemm_bf16_gpu.py tests/test_wgrad_gpu.py tests/test_lstm_gpu.py tests/test_config5_bf16_gpu.py tests/test_bf16_storage_gpu.py -q -m gpu -s 2>&1 | grep -v "^$" | tail -40) > $O/pytest.txt
cat $O/pytest.txt
python tools/gemm_bf16_time.py 2>&1 | grep -v amdgpu > $O/gemm_bf16.txt; cat $O/gemm_bf16.txt
for i in 1 2; do
  (python bench.py --dtype bf16 --no-alt --cpu-sample 0 --no-micro --graph on 2>/dev/null | tail -1 | cut -c1-200) >> $O/bench_bf16_own.txt
  (TSG_OWN_GEMM_BF16=0 python bench.py --dtype bf16 --no-alt --cpu-sample 0 --no-micro --graph on 2>/dev/null | tail -1 | cut -c1-200) >> $O/bench_bf16_lib.txt
done
echo "bf16 step, own GEMM:"; cat $O/bench_bf16_own.txt; echo "bf16 step, library GEMM:"; cat $O/bench_bf16_lib.txt
(python bench.py --no-alt --cpu-sample 0 --no-micro --graph on 2>/dev/null | tail -1 | cut -c1-200) > $O/bench_f32s.txt; cat $O/bench_f32s.txt
