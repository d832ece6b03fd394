#!/bin/bash
# round 5, session 24: bf16 storage backward, dG tile as 16-byte pieces: parity + A/B
O=gpurun_out/r5x; mkdir -p $O
(timeout 2400 python -m pytest tests/test_lstm_gpu.py tests/test_lstm_soak_gpu.py tests/test_bf16_storage_gpu.py tests/test_config5_bf16_gpu.py -q -m gpu -x 2>&1 | grep -v "^$" | tail -4) > $O/pytest.txt
cat $O/pytest.txt
run() { local label=$1; shift
  echo "== $label" >> $O/lstm_ab.txt
  env "$@" python -u tools/lstm_bench.py $SHAPE 2>&1 | grep -v amdgpu | grep "persistent backward" | sed 's/, err word.*//' >> $O/lstm_ab.txt
}
export TSG_BM=1 TSG_STEPK=0 TSG_REC_DTYPE=1
for rep in 1 2; do
  for SHAPE in "128 128 512" "64 128 512" "32 512 512" "64 20 512"; do
    run "bf16 [$SHAPE] four 2-byte dG stores per thread" TSG_HIP_LIB=tools/_ablate/prevlstm.so
    run "bf16 [$SHAPE] dG tile as 16-byte pieces" X=1
  done
done
cat $O/lstm_ab.txt
for rep in 1 2 3; do
for lib in shufflingvideosfortsg_amd/libtsg_hip.so tools/_ablate/prevlstm.so; do
      echo "lib=$lib bf16: $(TSG_HIP_LIB=$lib python bench.py --dtype bf16 --no-alt --cpu-sample 0 --no-micro --graph on 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])')" >> $O/bench.txt
done
done
sort $O/bench.txt
