#!/bin/bash
# K2 backward: parity (attention / model suites), then timing of the dS-once path vs the two full kernels (TSG_K2_BWD=pair), alternating processes.
#   usage: k2_bwd_ab.sh OUT
O=gpurun_out/$1; mkdir -p $O
(timeout 2400 python -m pytest tests/test_mha_gpu.py tests/test_models_gpu.py tests/test_config4_gpu.py tests/test_bf16_storage_gpu.py -q -m gpu 2>&1 | tail -12) > $O/pytest.txt
cat $O/pytest.txt
for rep in 1 2; do
  echo "== dS once" >> $O/k2.txt; python tools/k2_bwd_time.py 2>&1 | grep -v amdgpu >> $O/k2.txt
  echo "== pair (TSG_K2_BWD=pair)" >> $O/k2.txt; TSG_K2_BWD=pair python tools/k2_bwd_time.py 2>&1 | grep -v amdgpu >> $O/k2.txt
done
cat $O/k2.txt
