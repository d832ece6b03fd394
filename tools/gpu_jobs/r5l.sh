#!/bin/bash
# round 5, session 12: 64-unit bf16 forward, LDS tiles on conflict-free strides
O=gpurun_out/r5l; mkdir -p $O
(timeout 1500 python -m pytest tests/test_lstm_gpu.py tests/test_lstm_soak_gpu.py -q -m gpu 2>&1 | grep -v "^$" | tail -6) > $O/pytest.txt
cat $O/pytest.txt
run() { local label=$1; shift
  echo "== $label" >> $O/lstm_ab.txt
  env "$@" python -u tools/lstm_bench.py $SHAPE 2>&1 | grep -v amdgpu | grep "rec dtype" >> $O/lstm_ab.txt
}
export TSG_BM=1 TSG_STEPK=0 TSG_REC_DTYPE=1
for rep in 1 2; do
for SHAPE in "128 128 512" "256 128 512" "96 128 512"; do
  run "[$SHAPE] 32-unit ring kernel" TSG_LSTM_W64=0
  run "[$SHAPE] 64-unit" TSG_LSTM_W64=1
  run "[$SHAPE] 64-unit, neither stream" TSG_LSTM_W64=1 TSG_HIP_LIB=tools/_ablate/w64a3.so
done
done
cat $O/lstm_ab.txt
for i in 1 2; do
  (python bench.py --dtype bf16 --no-alt --cpu-sample 0 --no-micro --graph on 2>/dev/null | tail -1 | cut -c1-200) >> $O/bench_bf16_w64.txt
  (TSG_LSTM_W64=0 python bench.py --dtype bf16 --no-alt --cpu-sample 0 --no-micro --graph on 2>/dev/null | tail -1 | cut -c1-200) >> $O/bench_bf16_w32.txt
done
echo "bf16 step, 64-unit forward:"; cat $O/bench_bf16_w64.txt; echo "bf16 step, 32-unit forward:"; cat $O/bench_bf16_w32.txt
