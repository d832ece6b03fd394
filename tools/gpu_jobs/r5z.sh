#!/bin/bash
# round 5, session 26: K1g forward at N = 25 / 26 in fp32 storage on 8 + 8 waves (words beyond 24 as fp32 FMAs): parity + A/B against 4 + 4 (TSG_K1_PW=4)
O=gpurun_out/r5z; mkdir -p $O
(timeout 2400 python -m pytest tests/test_scdm_gpu.py tests/test_config4_gpu.py -q -m gpu -x 2>&1 | grep -v "^$" | tail -4) > $O/pytest.txt
cat $O/pytest.txt
for rep in 1 2; do
for shape in "64 256 25" "128 128 25" "64 256 26" "64 256 24" "128 128 20"; do
  echo "== [$shape, 1024] 4 + 4 waves (TSG_K1_PW=4)" >> $O/k1.txt; TSG_K1_PW=4 python tools/k1_fwd_modes_time.py $shape 2>/dev/null | grep "f32s" >> $O/k1.txt
  echo "== [$shape, 1024] default" >> $O/k1.txt; python tools/k1_fwd_modes_time.py $shape 2>/dev/null | grep "f32s" >> $O/k1.txt
done
done
cat $O/k1.txt
