#!/bin/bash
# round 5, session 18: backward, who issues what (TIMING-ONLY builds, -DTSG_BWD_ROLES bits: 1 streamed loads by waves 4-7 only, 2 poll by waves 0-3 only (8 pieces each),
# 4 partial stores by waves 0-3 only, 8 streamed loads one step ahead)
O=gpurun_out/r5r; mkdir -p $O
run() { local label=$1; shift
  echo "== $label" >> $O/lstm_ab.txt
  env "$@" python -u tools/lstm_bench.py $SHAPE 2>&1 | grep -v amdgpu | grep "persistent backward" | sed 's/, err word.*//' >> $O/lstm_ab.txt
}
export TSG_BM=1 TSG_STEPK=0
for dt in 2 1; do
  export TSG_REC_DTYPE=$dt
  for SHAPE in "128 128 512" "64 128 512"; do
    run "dt=$dt [$SHAPE] as shipped" X=1
    for r in 1 3 8 9 11 15; do run "dt=$dt [$SHAPE] roles=$r" TSG_HIP_LIB=tools/_ablate/br$r.so; done
  done
done
cat $O/lstm_ab.txt
