#!/bin/bash
# round 5, session 27: K1g forward, bf16 storage at N = 25: 26 instead of 28 score slots: parity + timing
O=gpurun_out/r5z2; mkdir -p $O
(timeout 2400 python -m pytest tests/test_scdm_gpu.py tests/test_config5_bf16_gpu.py tests/test_bf16_storage_gpu.py -q -m gpu -x 2>&1 | grep -v "^$" | grep "passed\|failed" | tail -4) > $O/pytest.txt
cat $O/pytest.txt
for rep in 1 2; do
for shape in "64 256 25" "128 512 25" "128 128 25"; do
  echo "== [$shape, 1024]" >> $O/k1.txt; python tools/k1_fwd_modes_time.py $shape 2>/dev/null | grep "bf16:" >> $O/k1.txt
done
done
cat $O/k1.txt
