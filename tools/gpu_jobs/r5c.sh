#!/bin/bash
# round 5, session 3: stream-K weight gradient (parity + A/B), LSTM ring policy, config-5 bf16 gradient norms, bench
O=gpurun_out/r5c; mkdir -p $O
(timeout 1200 python -m pytest tests/test_wgrad_gpu.py tests/test_lstm_gpu.py tests/test_config5_bf16_gpu.py -x -q -m gpu -s 2>&1 | grep -v "^$" | tail -25) > $O/pytest.txt
cat $O/pytest.txt
for rep in 1; do
  echo "== stream-K" >> $O/wgrad_ab.txt; python tools/wgrad_sk_time.py 2>&1 | grep -v amdgpu >> $O/wgrad_ab.txt
  echo "== split scheme (TSG_WGRAD_SK=0)" >> $O/wgrad_ab.txt; TSG_WGRAD_SK=0 python tools/wgrad_sk_time.py 2>&1 | grep -v amdgpu >> $O/wgrad_ab.txt
done
cat $O/wgrad_ab.txt
for i in 1; do
  (python bench.py --no-alt --cpu-sample 0 --no-micro --graph on 2>/dev/null | tail -1 | cut -c1-200) >> $O/bench.txt
  (TSG_WGRAD_SK=0 python bench.py --no-alt --cpu-sample 0 --no-micro --graph on 2>/dev/null | tail -1 | cut -c1-200) >> $O/bench_sk0.txt
done
echo "bench (stream-K):"; cat $O/bench.txt; echo "bench (split scheme):"; cat $O/bench_sk0.txt
