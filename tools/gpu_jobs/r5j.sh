#!/bin/bash
# round 5, session 10: why is the 64-unit kernel slower than its emulation at 128 rows?  ablations
O=gpurun_out/r5j; mkdir -p $O
run() { local label=$1; shift
  echo "== $label" >> $O/lstm_ab.txt
  env "$@" python -u tools/lstm_bench.py $SHAPE 2>&1 | grep -v amdgpu | grep "rec dtype" >> $O/lstm_ab.txt
}
export TSG_BM=1 TSG_STEPK=0 TSG_REC_DTYPE=1 TSG_LSTM_W64=1
for SHAPE in "128 128 512" "256 128 512"; do
  run "[$SHAPE] 64-unit as built" X=1
  run "[$SHAPE] no R / Cs stores" TSG_HIP_LIB=tools/_ablate/w64a1.so
  run "[$SHAPE] no Gx loads" TSG_HIP_LIB=tools/_ablate/w64a2.so
  run "[$SHAPE] neither" TSG_HIP_LIB=tools/_ablate/w64a3.so
  run "[$SHAPE] neither, no out store" TSG_HIP_LIB=tools/_ablate/w64a7.so
  run "[$SHAPE] time-major, as built" TSG_BM=0
done
cat $O/lstm_ab.txt
