#!/bin/bash
# round 5, session 17: what the backward step waits for -- streams compiled out one at a time (timing builds)
O=gpurun_out/r5q; mkdir -p $O
run() { local label=$1; shift
  echo "== $label" >> $O/lstm_ab.txt
  env "$@" python -u tools/lstm_bench.py $SHAPE 2>&1 | grep -v amdgpu | grep "persistent backward" | sed 's/, err word.*//' >> $O/lstm_ab.txt
}
export TSG_BM=1 TSG_STEPK=0
for dt in 2 1; do
  export TSG_REC_DTYPE=$dt
  for SHAPE in "128 128 512" "64 128 512"; do
    run "dt=$dt [$SHAPE] as shipped" X=1
    run "dt=$dt [$SHAPE] no dG stores" TSG_HIP_LIB=tools/_ablate/ba1.so
    run "dt=$dt [$SHAPE] no operand loads" TSG_HIP_LIB=tools/_ablate/ba2.so
    run "dt=$dt [$SHAPE] neither" TSG_HIP_LIB=tools/_ablate/ba3.so
    run "dt=$dt [$SHAPE] no c(t-1) load" TSG_HIP_LIB=tools/_ablate/ba4.so
  done
done
cat $O/lstm_ab.txt
