#!/bin/bash
# round 5, session 15: backward with its streamed operands requested one step ahead: parity + A/B; f32s / bf16 steps
O=gpurun_out/r5o; mkdir -p $O
(timeout 2400 python -m pytest tests/test_lstm_gpu.py tests/test_lstm_soak_gpu.py tests/test_config4_gpu.py tests/test_models_gpu.py tests/test_bf16_storage_gpu.py -q -m gpu 2>&1 | grep -v "^$" | tail -8) > $O/pytest.txt
cat $O/pytest.txt
run() { local label=$1; shift
  echo "== $label" >> $O/lstm_ab.txt
  env "$@" python -u tools/lstm_bench.py $SHAPE 2>&1 | grep -v amdgpu | grep "persistent backward" | sed 's/, err word.*//' >> $O/lstm_ab.txt
}
export TSG_BM=1 TSG_STEPK=0
for rep in 1 2; do
for dt in 2 1; do
  export TSG_REC_DTYPE=$dt
  for SHAPE in "128 128 512" "64 128 512" "32 512 512" "64 20 512"; do
    run "dt=$dt [$SHAPE] operands in front of the poll (r4)" TSG_HIP_LIB=tools/_ablate/bwdna.so
    run "dt=$dt [$SHAPE] operands one step ahead" X=1
  done
done
done
cat $O/lstm_ab.txt
for i in 1 2 3; do
  (python bench.py --no-alt --cpu-sample 0 --no-micro --graph on 2>/dev/null | tail -1 | cut -c1-200) >> $O/bench_f32s.txt
  (TSG_HIP_LIB=tools/_ablate/prevlstm.so python bench.py --no-alt --cpu-sample 0 --no-micro --graph on 2>/dev/null | tail -1 | cut -c1-200) >> $O/bench_f32s_prev.txt
done
for i in 1 2; do
  (python bench.py --dtype bf16 --no-alt --cpu-sample 0 --no-micro --graph on 2>/dev/null | tail -1 | cut -c1-200) >> $O/bench_bf16.txt
done
echo "f32s step now:"; cat $O/bench_f32s.txt; echo "f32s step, lstm.hip of two commits ago (per-lane streams, operands in front of the poll):"; cat $O/bench_f32s_prev.txt; echo "bf16 step now:"; cat $O/bench_bf16.txt
