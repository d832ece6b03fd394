#!/bin/bash
# K1g forward A/B on one box: parity of the tree's library, then stand-alone timings of the tree's library against
# variant libraries (tools/_ablate/<name>.so, TSG_HIP_LIB), alternating processes; then ticks of instrumented builds.
#   usage: k1_fwd_ab.sh OUTDIR "variants" "tick libs"
O=gpurun_out/$1; mkdir -p $O
(timeout 1800 python -m pytest tests/test_scdm_gpu.py tests/test_bf16_storage_gpu.py -q -m gpu -x 2>&1 | grep "passed\|failed\|Error\|error" | tail -6) > $O/pytest.txt
cat $O/pytest.txt
for rep in 1 2; do
  echo "== tree" >> $O/k1.txt; python tools/k1_variants.py 128 200 2>&1 | grep "^f32s\|^bf16" >> $O/k1.txt
  for v in $2; do
    echo "== $v" >> $O/k1.txt; TSG_HIP_LIB=tools/_ablate/$v.so python tools/k1_variants.py 128 200 2>&1 | grep "^f32s\|^bf16" >> $O/k1.txt
  done
done
for v in $3; do
  echo "=== $v gate=1" >> $O/k1.txt
  TSG_HIP_LIB=tools/_ablate/$v.so python tools/k1_ticks.py 128 1 2 2>&1 | grep -v amdgpu.ids | head -34 >> $O/k1.txt
done
cat $O/k1.txt
