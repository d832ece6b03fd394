#!/bin/bash
# round 5, session 32: forward Gx tile by non-temporal LDS-DMA (fp32 storage): parity + A/B against the register prefetch
O=gpurun_out/r5gx; mkdir -p $O
(timeout 2400 python -m pytest tests/test_lstm_gpu.py tests/test_lstm_soak_gpu.py tests/test_config4_gpu.py tests/test_models_gpu.py -q -m gpu -x 2>&1 | grep "passed\|failed\|^E " | head -6) > $O/pytest.txt; cat $O/pytest.txt
run() { local label=$1; shift
  echo "== $label" >> $O/lstm_ab.txt
  env "$@" python -u tools/lstm_bench.py $SHAPE 2>&1 | grep -v amdgpu | grep "rec dtype" | sed 's/rec dtype.*: fwd/   fwd/; s/, bwd 0.00.*//' >> $O/lstm_ab.txt
}
export TSG_BM=1 TSG_STEPK=0
for rep in 1 2; do
for dt in 2 0; do
  export TSG_REC_DTYPE=$dt
  for SHAPE in "128 128 512" "64 128 512" "32 512 512" "64 20 512"; do
    run "dt=$dt [$SHAPE] Gx piece in registers over the step" TSG_HIP_LIB=tools/_ablate/nogxd.so
    run "dt=$dt [$SHAPE] Gx tile by LDS-DMA" X=1
  done
done
done
cat $O/lstm_ab.txt
for rep in 1 2 3; do
for lib in shufflingvideosfortsg_amd/libtsg_hip.so tools/_ablate/nogxd.so; do
      echo "lib=$lib f32s: $(TSG_HIP_LIB=$lib python bench.py --no-alt --cpu-sample 0 --no-micro --graph on 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])')" >> $O/bench.txt
done
done
sort $O/bench.txt
