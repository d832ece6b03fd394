#!/bin/bash
# round 5, session 8: would 64-unit workgroups on half the CUs pay in bf16 storage?  Timing-only emulation: the 128-workgroup grid of 64 rows
# (the L2 traffic and the clock of the half-chip design) with each wave's MFMAs and gates DOUBLED (-DTSG_LSTM_DOUBLE_WORK)
O=gpurun_out/r5h; mkdir -p $O
run() { local label=$1; shift
  echo "== $label" >> $O/lstm_ab.txt
  env "$@" python -u tools/lstm_bench.py $SHAPE 2>&1 | grep -v amdgpu | grep "rec dtype\|sync word0" >> $O/lstm_ab.txt
}
export TSG_BM=1 TSG_STEPK=0 TSG_REC_DTYPE=1
for rep in 1 2; do
  SHAPE="128 128 512"; run "bf16 [128 rows] as shipped (ring)" TSG_RING=1
  SHAPE="64 128 512"
  run "bf16 [64 rows] out-polling" TSG_RING=0 TSG_LSTM_XR=0
  run "bf16 [64 rows] ring" TSG_RING=1 TSG_LSTM_XR=1
  run "bf16 [64 rows] out-polling, DOUBLE work" TSG_RING=0 TSG_LSTM_XR=0 TSG_HIP_LIB=tools/_ablate/dw.so
  run "bf16 [64 rows] ring, DOUBLE work" TSG_RING=1 TSG_LSTM_XR=1 TSG_HIP_LIB=tools/_ablate/dw.so
done
SHAPE="64 128 512"
run "bf16 [64 rows] out-polling, DOUBLE work, phase ticks" TSG_RING=0 TSG_LSTM_XR=0 TSG_HIP_LIB=tools/_ablate/dwt.so
run "bf16 [64 rows] ring, DOUBLE work, phase ticks" TSG_RING=1 TSG_LSTM_XR=1 TSG_HIP_LIB=tools/_ablate/dwt.so
cat $O/lstm_ab.txt
