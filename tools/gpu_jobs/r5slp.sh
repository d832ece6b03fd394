#!/bin/bash
# round 5, session 31: lstm.hip / scdm_attn.hip built without the SLP vectoriser (packed fp32 VALU beside MFMAs): A/B
O=gpurun_out/r5slp; mkdir -p $O
run() { local label=$1; shift
  echo "== $label" >> $O/lstm_ab.txt
  env "$@" python -u tools/lstm_bench.py $SHAPE 2>&1 | grep -v amdgpu | grep "persistent backward\|rec dtype" | sed 's/, err word.*//; s/rec dtype.*: fwd/   fwd/; s/, bwd 0.00.*//' >> $O/lstm_ab.txt
}
export TSG_BM=1 TSG_STEPK=0
for rep in 1 2; do
for dt in 2 1; do
  export TSG_REC_DTYPE=$dt
  for SHAPE in "128 128 512"; do
    run "dt=$dt [$SHAPE] default" X=1
    run "dt=$dt [$SHAPE] lstm.hip -fno-slp-vectorize" TSG_HIP_LIB=tools/_ablate/noslp.so
  done
done
done
cat $O/lstm_ab.txt
for rep in 1 2; do
for shape in "128 128 20"; do
  echo "== K1g fwd default" >> $O/k1.txt; python tools/k1_fwd_modes_time.py $shape 2>/dev/null | grep "f32s\|bf16:" >> $O/k1.txt
  echo "== K1g fwd scdm_attn.hip -fno-slp-vectorize" >> $O/k1.txt; TSG_HIP_LIB=tools/_ablate/noslpk1.so python tools/k1_fwd_modes_time.py $shape 2>/dev/null | grep "f32s\|bf16:" >> $O/k1.txt
done
done
cat $O/k1.txt
for rep in 1 2 3; do
for lib in shufflingvideosfortsg_amd/libtsg_hip.so tools/_ablate/noslp.so tools/_ablate/noslpk1.so; do
      echo "lib=$lib f32s: $(TSG_HIP_LIB=$lib python bench.py --no-alt --cpu-sample 0 --no-micro --graph on 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])')" >> $O/bench.txt
done
done
sort $O/bench.txt
