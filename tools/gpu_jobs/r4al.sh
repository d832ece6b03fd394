#!/bin/bash
mkdir -p gpurun_out/r4al
for v in base wabl1 wabl2 wabl4 wabl8; do
  echo "== $v (TSG_WGRAD_ABL: 1 no barrier, 2 no loads, 4 no staging, 8 no MFMAs)"
  if [ $v = base ]; then python tools/wgrad_bf16_time.py 2>&1 | grep -v amdgpu | tail -2
  else TSG_HIP_LIB=$PWD/tools/_ablate/$v.so python tools/wgrad_bf16_time.py 2>&1 | grep -v amdgpu | tail -2; fi
done > gpurun_out/r4al/abl.txt
cat gpurun_out/r4al/abl.txt
