#!/bin/bash
# round 5, session 14: f32s forward with staged streams: where is the rest?  stream ablations + phase ticks; bf16 step with the new policy
O=gpurun_out/r5n; mkdir -p $O
run() { local label=$1; shift
  echo "== $label" >> $O/lstm_ab.txt
  env "$@" python -u tools/lstm_bench.py $SHAPE 2>&1 | grep -v amdgpu | grep "rec dtype\|sync word0" | sed 's/persistent backward.*//' >> $O/lstm_ab.txt
}
export TSG_BM=1 TSG_STEPK=0
for dt in 2 1; do
  export TSG_REC_DTYPE=$dt
  for SHAPE in "128 128 512" "64 128 512"; do
    run "dt=$dt [$SHAPE] staged streams" X=1
    run "dt=$dt [$SHAPE] no R / Cs stores" TSG_HIP_LIB=tools/_ablate/fa1.so
    run "dt=$dt [$SHAPE] no Gx loads" TSG_HIP_LIB=tools/_ablate/fa2.so
    run "dt=$dt [$SHAPE] no streams at all" TSG_HIP_LIB=tools/_ablate/fa7.so
    run "dt=$dt [$SHAPE] phase ticks" TSG_HIP_LIB=tools/_ablate/ftim.so
    run "dt=$dt [$SHAPE] out-polling (no ring)" TSG_RING=0
  done
done
cat $O/lstm_ab.txt
for i in 1 2; do
  (python bench.py --dtype bf16 --no-alt --cpu-sample 0 --no-micro --graph on 2>/dev/null | tail -1 | cut -c1-200) >> $O/bench_bf16.txt
  (TSG_HIP_LIB=tools/_ablate/prevlstm.so TSG_LSTM_W64=0 python bench.py --dtype bf16 --no-alt --cpu-sample 0 --no-micro --graph on 2>/dev/null | tail -1 | cut -c1-200) >> $O/bench_bf16_prev.txt
done
echo "bf16 step, staged streams:"; cat $O/bench_bf16.txt; echo "bf16 step, per-lane streams, 32-unit:"; cat $O/bench_bf16_prev.txt
