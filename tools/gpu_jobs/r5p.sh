#!/bin/bash
# round 5, session 16: backward with its streamed loads BEHIND the poll loads (counted wait): parity + A/B
O=gpurun_out/r5p; mkdir -p $O
(timeout 2400 python -m pytest tests/test_lstm_gpu.py tests/test_lstm_soak_gpu.py tests/test_config4_gpu.py tests/test_models_gpu.py tests/test_bf16_storage_gpu.py -q -m gpu 2>&1 | grep -v "^$" | tail -6) > $O/pytest.txt
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
    run "dt=$dt [$SHAPE] operands in front of the poll (r4)" TSG_HIP_LIB=tools/_ablate/prevlstm.so
    run "dt=$dt [$SHAPE] operands behind the poll loads, counted wait" X=1
  done
done
done
cat $O/lstm_ab.txt
for i in 1 2 3; do
  (python bench.py --no-alt --cpu-sample 0 --no-micro --graph on 2>/dev/null | tail -1 | cut -c1-200) >> $O/bench_f32s.txt
  (TSG_HIP_LIB=tools/_ablate/prevlstm.so python bench.py --no-alt --cpu-sample 0 --no-micro --graph on 2>/dev/null | tail -1 | cut -c1-200) >> $O/bench_f32s_prev.txt
done
echo "f32s step now:"; cat $O/bench_f32s.txt; echo "f32s step, lstm.hip of three commits ago:"; cat $O/bench_f32s_prev.txt
