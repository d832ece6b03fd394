#!/bin/bash
# round 5, session 2: phase ticks (shader cycles) of the forward at 128 / 16 rows, ring on / off; bf16 storage A/B
O=gpurun_out/r5b; mkdir -p $O
run() { local label=$1; shift
  echo "== $label" >> $O/lstm_ab.txt
  env "$@" python -u tools/lstm_bench.py $SHAPE 2>&1 | grep -v amdgpu | grep "rec dtype\|persistent backward\|sync word0" >> $O/lstm_ab.txt
}
export TSG_BM=1 TSG_STEPK=0
for dt in 2 1; do
  export TSG_REC_DTYPE=$dt
  for SHAPE in "128 128 512" "16 128 512" "64 128 512"; do
    run "timing build dt=$dt $SHAPE out-polling" TSG_RING=0 TSG_HIP_LIB=tools/_ablate/timing.so
    run "timing build dt=$dt $SHAPE ring" TSG_RING=1 TSG_HIP_LIB=tools/_ablate/timing.so
  done
done
export TSG_REC_DTYPE=1
for rep in 1 2; do
  SHAPE="128 128 512"
  run "dt=1 out-polling" TSG_RING=0
  run "dt=1 ring" TSG_RING=1
  run "dt=1 ring NW=4" TSG_RING=1 TSG_LSTM_NW=4
  SHAPE="32 512 512"
  run "dt=1 $SHAPE out-polling" TSG_RING=0
  run "dt=1 $SHAPE ring" TSG_RING=1
  run "dt=1 $SHAPE ring NW=8" TSG_RING=1 TSG_LSTM_NW=8
done
cat $O/lstm_ab.txt
