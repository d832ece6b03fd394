#!/bin/bash
# round 5, session 21: backward operand tiles by LDS-DMA: one vs two steps of look-ahead
O=gpurun_out/r5u; mkdir -p $O
run() { local label=$1; shift
  echo "== $label" >> $O/lstm_ab.txt
  env "$@" python -u tools/lstm_bench.py $SHAPE 2>&1 | grep -v amdgpu | grep "persistent backward" | sed 's/, err word.*//' >> $O/lstm_ab.txt
}
export TSG_BM=1 TSG_STEPK=0
for rep in 1 2; do
for dt in 2 1; do
  export TSG_REC_DTYPE=$dt
  for SHAPE in "128 128 512" "64 128 512"; do
    run "dt=$dt [$SHAPE] operands in front of the poll (r4)" TSG_HIP_LIB=tools/_ablate/prevlstm.so
    run "dt=$dt [$SHAPE] DMA one step ahead, behind the poll barrier" X=1
    run "dt=$dt [$SHAPE] DMA one step ahead, behind the dG tile barrier" TSG_HIP_LIB=tools/_ablate/pos1.so
    run "dt=$dt [$SHAPE] DMA two steps ahead, behind the poll barrier" TSG_HIP_LIB=tools/_ablate/ah2.so
    run "dt=$dt [$SHAPE] DMA two steps ahead, behind the dG tile barrier" TSG_HIP_LIB=tools/_ablate/ah2p1.so
  done
done
done
cat $O/lstm_ab.txt
