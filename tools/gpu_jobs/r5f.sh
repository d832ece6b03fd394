#!/bin/bash
# round 5, session 6: own Adam (parity, A/B in the step), gated DP at world 1, config-5 bounds
O=gpurun_out/r5f; mkdir -p $O
(timeout 1500 python -m pytest tests/test_adam_gpu.py tests/test_dp_rccl_gpu.py tests/test_bench_gpu.py tests/test_config5_bf16_gpu.py tests/test_models_gpu.py -q -m gpu 2>&1 | grep -v "^$" | tail -25) > $O/pytest.txt
cat $O/pytest.txt
for i in 1 2 3; do
  (python bench.py --no-alt --cpu-sample 0 --no-micro --graph on 2>/dev/null | tail -1 | cut -c1-200) >> $O/bench_own_adam.txt
  (TSG_OWN_ADAM=0 python bench.py --no-alt --cpu-sample 0 --no-micro --graph on 2>/dev/null | tail -1 | cut -c1-200) >> $O/bench_torch_adam.txt
done
echo "own Adam:"; cat $O/bench_own_adam.txt; echo "torch fused Adam:"; cat $O/bench_torch_adam.txt
