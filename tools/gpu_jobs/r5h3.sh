#!/bin/bash
# round 5, session 29: gradient sinks for the shared activations: parity + step A/B
O=gpurun_out/r5h3; mkdir -p $O
(timeout 2400 python -m pytest tests/test_models_gpu.py tests/test_fullsize_gpu.py tests/test_config4_gpu.py tests/test_head_gemm_gpu.py tests/test_match_head_gpu.py tests/test_scdm_gpu.py -q -m gpu 2>&1 | grep -v "^$" | tail -8) > $O/pytest.txt
cat $O/pytest.txt
for rep in; do
  echo "sinks:    $(python bench.py --no-alt --cpu-sample 0 --no-micro --graph on 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"], d["eager"]["ms_per_step"])')" >> $O/bench.txt
  echo "autograd: $(TSG_SHARED_GRAD=0 python bench.py --no-alt --cpu-sample 0 --no-micro --graph on 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"], d["eager"]["ms_per_step"])')" >> $O/bench.txt
done
cat $O/bench.txt
