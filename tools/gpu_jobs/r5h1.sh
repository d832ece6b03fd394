#!/bin/bash
# round 5, session 28: recalibration tail as one autograd node (dx accumulated in the GEMM epilogue): parity + step A/B
O=gpurun_out/r5h1; mkdir -p $O
(timeout 2400 python -m pytest tests/test_scdm_gpu.py tests/test_models_gpu.py tests/test_fullsize_gpu.py tests/test_gemm_f32s_gpu.py -q -m gpu -x 2>&1 | grep -v "^$" | tail -6) > $O/pytest.txt
cat $O/pytest.txt
for rep in 1 2 3; do
  echo "one node:  $(python bench.py --no-alt --cpu-sample 0 --no-micro --graph on 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"], d["eager"]["ms_per_step"])')" >> $O/bench.txt
  echo "two nodes: $(TSG_SHARED_GRAD=0 python bench.py --no-alt --cpu-sample 0 --no-micro --graph on 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"], d["eager"]["ms_per_step"])')" >> $O/bench.txt
done
cat $O/bench.txt
