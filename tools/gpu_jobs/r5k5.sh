#!/bin/bash
# round 5, session 30: K5 backward with column-sliced workgroups: parity + A/B
O=gpurun_out/r5k5; mkdir -p $O
(timeout 1200 python -m pytest tests/test_match_head_gpu.py tests/test_head_gemm_gpu.py tests/test_bf16_storage_gpu.py tests/test_fullsize_gpu.py -q -m gpu 2>&1 | grep "passed\|failed\|^E " | head -8) > $O/pytest.txt; cat $O/pytest.txt
for rep in 1 2 3; do
  echo "(pair, 32 clips) workgroups: $(TSG_MH_BWD_COLS=0 python tools/k5_time.py 2>/dev/null | tail -1)" >> $O/k5.txt
  echo "column-sliced workgroups:    $(python tools/k5_time.py 2>/dev/null | tail -1)" >> $O/k5.txt
done
cat $O/k5.txt
for rep in 1 2; do
  for v in 1 0; do
    for dtype in f32s bf16; do
      echo "cols=$v $dtype: $(TSG_MH_BWD_COLS=$v python bench.py --dtype $dtype --no-alt --cpu-sample 0 --no-micro --graph on 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])')" >> $O/bench.txt
    done
  done
done
sort $O/bench.txt
