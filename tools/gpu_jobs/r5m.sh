#!/bin/bash
# round 5, session 13: the general forward kernel with Gx / R / Cs staged through LDS: parity (all LSTM tests + models), per-step A/B vs the per-lane streams, step A/B
O=gpurun_out/r5m; mkdir -p $O
(timeout 2400 python -m pytest tests/test_lstm_gpu.py tests/test_lstm_soak_gpu.py tests/test_split_gemm_gpu.py tests/test_config4_gpu.py tests/test_models_gpu.py tests/test_bf16_storage_gpu.py tests/test_config1_gpu.py -q -m gpu 2>&1 | grep -v "^$" | tail -12) > $O/pytest.txt
cat $O/pytest.txt
run() { local label=$1; shift
  echo "== $label" >> $O/lstm_ab.txt
  env "$@" python -u tools/lstm_bench.py $SHAPE 2>&1 | grep -v amdgpu | grep "rec dtype" >> $O/lstm_ab.txt
}
export TSG_BM=1 TSG_STEPK=0
for rep in 1 2; do
for dt in 2 1 0; do
  export TSG_REC_DTYPE=$dt
  for SHAPE in "128 128 512" "64 128 512" "32 512 512" "64 20 512"; do
    run "dt=$dt [$SHAPE] per-lane streams (previous)" TSG_HIP_LIB=tools/_ablate/prevlstm.so TSG_LSTM_W64=0
    run "dt=$dt [$SHAPE] staged streams" TSG_LSTM_W64=0
  done
done
done
cat $O/lstm_ab.txt
for i in 1 2 3; do
  (python bench.py --no-alt --cpu-sample 0 --no-micro --graph on 2>/dev/null | tail -1 | cut -c1-200) >> $O/bench_staged.txt
  (TSG_HIP_LIB=tools/_ablate/prevlstm.so python bench.py --no-alt --cpu-sample 0 --no-micro --graph on 2>/dev/null | tail -1 | cut -c1-200) >> $O/bench_prev.txt
done
echo "f32s step, staged streams:"; cat $O/bench_staged.txt; echo "f32s step, per-lane streams (previous commit's lstm.hip):"; cat $O/bench_prev.txt
