#!/bin/bash
mkdir -p gpurun_out/r4ac
for rep in 1 2; do
for v in base gemm_abl1 gemm_abl2; do
  echo "== $v"
  if [ $v = base ]; then python tools/gemm_f32s_time.py 2>&1 | grep -v amdgpu | cut -c1-95 | head -4
  else TSG_HIP_LIB=$PWD/tools/_ablate/$v.so python tools/gemm_f32s_time.py 2>&1 | grep -v amdgpu | cut -c1-95 | head -4; fi
done
done > gpurun_out/r4ac/abl.txt
cat gpurun_out/r4ac/abl.txt
