#!/bin/bash
# round 5, session 36: gradient sinks in the bf16 storage mode: parity + A/B
O=gpurun_out/r5bs; mkdir -p $O
(timeout 2400 python -m pytest tests/test_bf16_storage_gpu.py tests/test_config5_bf16_gpu.py tests/test_models_gpu.py tests/test_scdm_gpu.py tests/test_fullsize_gpu.py -q -m gpu 2>&1 | grep "passed\|failed\|^E " | head -8) > $O/pytest.txt; cat $O/pytest.txt
for rep in 1 2 3; do
  for v in 1 0; do
    echo "sinks=$v bf16: $(TSG_SHARED_GRAD=$v python bench.py --dtype bf16 --no-alt --cpu-sample 0 --no-micro --graph on 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"], d["eager"]["ms_per_step"])')" >> $O/bench.txt
  done
done
sort $O/bench.txt
