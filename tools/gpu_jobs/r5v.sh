#!/bin/bash
# round 5, session 22: non-temporal hints on the LSTM streams (backward DMA; forward Gx loads / R, Cs stores), stand-alone and in the train step
O=gpurun_out/r5v; mkdir -p $O
run() { local label=$1; shift
  echo "== $label" >> $O/lstm_ab.txt
  env "$@" python -u tools/lstm_bench.py $SHAPE 2>&1 | grep -v amdgpu | grep "persistent backward\|rec dtype" | sed 's/, err word.*//; s/rec dtype.*: fwd/   fwd/; s/, bwd 0.00.*//' >> $O/lstm_ab.txt
}
export TSG_BM=1 TSG_STEPK=0
for rep in 1 2; do
for dt in 2 1; do
  export TSG_REC_DTYPE=$dt
  for SHAPE in "128 128 512" "64 128 512"; do
    run "dt=$dt [$SHAPE] default" X=1
    run "dt=$dt [$SHAPE] backward DMA nt" TSG_HIP_LIB=tools/_ablate/bnt.so
    run "dt=$dt [$SHAPE] forward Gx nt" TSG_HIP_LIB=tools/_ablate/fnt1.so
    run "dt=$dt [$SHAPE] forward R/Cs nt" TSG_HIP_LIB=tools/_ablate/fnt2.so
    run "dt=$dt [$SHAPE] forward Gx + R/Cs nt" TSG_HIP_LIB=tools/_ablate/fnt3.so
  done
done
done
cat $O/lstm_ab.txt
for lib in "" tools/_ablate/bnt.so tools/_ablate/fnt3.so tools/_ablate/prevlstm.so; do
  for dtype in f32s bf16; do
    for i in 1 2; do
      echo "lib=${lib:-default} $dtype: $(TSG_HIP_LIB=$lib python bench.py --dtype $dtype --no-alt --cpu-sample 0 --no-micro --graph on 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])')" >> $O/bench.txt
    done
  done
done
cat $O/bench.txt
