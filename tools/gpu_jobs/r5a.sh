#!/bin/bash
# round 5, session 1: the forward's exchange ring (XR) + deferred R/Cs stores + backward touch-ahead / deferred dG: parity, then A/B timings
O=gpurun_out/r5a; mkdir -p $O
(timeout 900 python -m pytest tests/test_lstm_gpu.py -x -q -m gpu 2>&1 | tail -5) > $O/pytest_lstm.txt
cat $O/pytest_lstm.txt
run() { # label, env..., shape
  local label=$1; shift
  echo "== $label" >> $O/lstm_ab.txt
  env "$@" python tools/lstm_bench.py $SHAPE 2>&1 | grep -v amdgpu | grep "rec dtype\|persistent backward\|sync word0" | sed 's/phase ticks.*//' >> $O/lstm_ab.txt
}
for rep in 1 2; do
for dt in 2 1; do
  export TSG_REC_DTYPE=$dt TSG_BM=1
  SHAPE="128 128 512"
  run "dt=$dt out-polling (r4)" TSG_RING=0
  run "dt=$dt ring" TSG_RING=1
  run "dt=$dt ring NW=4" TSG_RING=1 TSG_LSTM_NW=4
  run "dt=$dt ring + deferred R/Cs" TSG_RING=1 TSG_HIP_LIB=tools/_ablate/defer.so
  run "dt=$dt out-polling + deferred R/Cs" TSG_RING=0 TSG_HIP_LIB=tools/_ablate/defer.so
  run "dt=$dt bwd touch" TSG_RING=1 TSG_HIP_LIB=tools/_ablate/bwdt.so
  run "dt=$dt bwd deferred dG" TSG_RING=1 TSG_HIP_LIB=tools/_ablate/bwdd.so
  run "dt=$dt all" TSG_RING=1 TSG_HIP_LIB=tools/_ablate/all.so
  run "dt=$dt all NW=4" TSG_RING=1 TSG_LSTM_NW=4 TSG_HIP_LIB=tools/_ablate/all.so
done
done
export TSG_REC_DTYPE=2 TSG_BM=1
for SHAPE in "32 512 512" "16 512 512" "64 20 512"; do
  run "dt=2 $SHAPE out-polling" TSG_RING=0
  run "dt=2 $SHAPE ring" TSG_RING=1
  run "dt=2 $SHAPE ring NW=8" TSG_RING=1 TSG_LSTM_NW=8
  run "dt=2 $SHAPE all" TSG_RING=1 TSG_HIP_LIB=tools/_ablate/all.so
done
export TSG_BM=0; SHAPE="128 128 512"
run "dt=2 time-major out-polling" TSG_RING=0
run "dt=2 time-major ring" TSG_RING=1
cat $O/lstm_ab.txt
