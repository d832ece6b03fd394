#!/bin/bash
# GPU job of round 4 (p): tsg_gemm_f32s experiments: scheduling fence behind the chunk barrier, 8 x 4 tile groups per XCD
mkdir -p gpurun_out/r4p
for rep in 1 2; do
for v in base gemm_sb gemm_xmap gemm_sbxmap; do
  echo "== $v" >> gpurun_out/r4p/gemm.txt
  if [ $v = base ]; then python tools/gemm_f32s_time.py 2>&1 | grep -v amdgpu | cut -c1-120 >> gpurun_out/r4p/gemm.txt
  else TSG_HIP_LIB=$PWD/tools/_ablate/$v.so python tools/gemm_f32s_time.py 2>&1 | grep -v amdgpu | cut -c1-120 >> gpurun_out/r4p/gemm.txt; fi
done
done
cat gpurun_out/r4p/gemm.txt
