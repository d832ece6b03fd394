#!/bin/bash
# K1 parity suites, then the tree's library vs one variant library, alternating processes, also on zero operands.   usage: k1_quick_ab.sh OUT variant
O=gpurun_out/$1; mkdir -p $O
(timeout 1800 python -m pytest tests/test_scdm_gpu.py tests/test_bf16_storage_gpu.py tests/test_config_anet256_gpu.py -q -m gpu 2>&1 | tail -3) | tee $O/pytest.txt
for rep in 1 2 3; do
  echo "== tree" >> $O/k1.txt; python tools/k1_variants.py 128 200 2>&1 | grep "^f32s\|^bf16" | head -2 >> $O/k1.txt
  echo "== $2" >> $O/k1.txt; TSG_HIP_LIB=tools/_ablate/$2.so python tools/k1_variants.py 128 200 2>&1 | grep "^f32s\|^bf16" | head -2 >> $O/k1.txt
done
paste - - - < $O/k1.txt | cut -c1-230
