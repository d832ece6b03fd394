#!/bin/bash
# round 5, session 9: the 64-unit bf16 forward kernel: parity, poisoned loop, timings, bf16 step A/B
O=gpurun_out/r5i; mkdir -p $O
(timeout 1500 python -m pytest tests/test_lstm_gpu.py tests/test_lstm_soak_gpu.py tests/test_bf16_storage_gpu.py tests/test_config5_bf16_gpu.py -q -m gpu 2>&1 | grep -v "^$" | tail -12) > $O/pytest.txt
cat $O/pytest.txt
run() { local label=$1; shift
  echo "== $label" >> $O/lstm_ab.txt
  env "$@" python -u tools/lstm_bench.py $SHAPE 2>&1 | grep -v amdgpu | grep "rec dtype\|sync word0" | sed 's/phase ticks.*//' >> $O/lstm_ab.txt
}
export TSG_BM=1 TSG_STEPK=0 TSG_REC_DTYPE=1
for rep in 1 2; do
  for SHAPE in "128 128 512" "256 128 512" "96 128 512"; do
    run "bf16 [$SHAPE] 32-unit ring kernel" TSG_LSTM_W64=0
    run "bf16 [$SHAPE] 64-unit kernel" TSG_LSTM_W64=1
  done
done
SHAPE="64 128 512"; run "bf16 [$SHAPE] 32-unit" TSG_LSTM_W64=0 TSG_LSTM_XR=1; run "bf16 [$SHAPE] 64-unit" TSG_LSTM_W64=1
cat $O/lstm_ab.txt
for i in 1 2; do
  (python bench.py --dtype bf16 --no-alt --cpu-sample 0 --no-micro --graph on 2>/dev/null | tail -1 | cut -c1-200) >> $O/bench_bf16_w64.txt
  (TSG_LSTM_W64=0 python bench.py --dtype bf16 --no-alt --cpu-sample 0 --no-micro --graph on 2>/dev/null | tail -1 | cut -c1-200) >> $O/bench_bf16_w32.txt
done
echo "bf16 step, 64-unit forward:"; cat $O/bench_bf16_w64.txt; echo "bf16 step, 32-unit forward:"; cat $O/bench_bf16_w32.txt
