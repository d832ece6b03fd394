#!/bin/bash
# round 5, session 20: backward operand tiles by LDS-DMA one step ahead: parity (LSTM suites) + A/B against the loads in front of the poll
O=gpurun_out/r5t; mkdir -p $O
(timeout 2400 python -m pytest tests/test_lstm_gpu.py tests/test_lstm_soak_gpu.py tests/test_config4_gpu.py tests/test_models_gpu.py tests/test_bf16_storage_gpu.py -q -m gpu -x 2>&1 | grep -v "^$" | tail -12) > $O/pytest.txt
cat $O/pytest.txt
run() { local label=$1; shift
  echo "== $label" >> $O/lstm_ab.txt
  env "$@" python -u tools/lstm_bench.py $SHAPE 2>&1 | grep -v amdgpu | grep "persistent backward" | sed 's/, err word.*//' >> $O/lstm_ab.txt
}
export TSG_BM=1 TSG_STEPK=0
for rep in 1 2; do
for dt in 2 1 0; do
  export TSG_REC_DTYPE=$dt
  for SHAPE in "128 128 512" "64 128 512" "32 512 512" "64 20 512"; do
    run "dt=$dt [$SHAPE] operands in front of the poll (r4)" TSG_HIP_LIB=tools/_ablate/prevlstm.so
    run "dt=$dt [$SHAPE] operand tiles by LDS-DMA, requested behind the poll barrier" X=1
    run "dt=$dt [$SHAPE] operand tiles by LDS-DMA, requested behind the dG tile barrier" TSG_HIP_LIB=tools/_ablate/pos1.so
  done
done
done
cat $O/lstm_ab.txt
