#!/bin/bash
# round 5, session 34: selective poll retries (only the pieces that were not ready are requested again): parity + A/B
O=gpurun_out/r5sel; mkdir -p $O
(timeout 2400 python -m pytest tests/test_lstm_gpu.py tests/test_lstm_soak_gpu.py tests/test_config4_gpu.py -q -m gpu -x 2>&1 | grep "passed\|failed\|^E " | head -6) > $O/pytest.txt; cat $O/pytest.txt
run() { local label=$1; shift
  echo "== $label" >> $O/lstm_ab.txt
  env "$@" python -u tools/lstm_bench.py $SHAPE 2>&1 | grep -v amdgpu | grep "persistent backward\|rec dtype" | sed 's/, err word.*//; s/rec dtype.*: fwd/   fwd/; s/, bwd 0.00.*//' >> $O/lstm_ab.txt
}
export TSG_BM=1 TSG_STEPK=0
for rep in 1 2; do
for dt in 2 1; do
  export TSG_REC_DTYPE=$dt
  for SHAPE in "128 128 512" "64 128 512" "32 512 512"; do
    run "dt=$dt [$SHAPE] whole slab again on a retry" TSG_HIP_LIB=tools/_ablate/nosel.so
    run "dt=$dt [$SHAPE] selective retries" X=1
  done
done
done
cat $O/lstm_ab.txt
for rep in 1 2 3; do
for lib in shufflingvideosfortsg_amd/libtsg_hip.so tools/_ablate/nosel.so; do
  for dtype in f32s bf16; do
      echo "lib=$lib $dtype: $(TSG_HIP_LIB=$lib python bench.py --dtype $dtype --no-alt --cpu-sample 0 --no-micro --graph on 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])')" >> $O/bench.txt
  done
done
done
sort $O/bench.txt
