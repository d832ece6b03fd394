#!/bin/bash
# GPU job of round 4 (q): -fno-slp-vectorize (no v_pk_add_f32 beside the MFMAs) in the two GEMM kernels
mkdir -p gpurun_out/r4q
for rep in 1 2; do
for v in gemm_sbxmap gemm_noslp; do
  echo "== $v" >> gpurun_out/r4q/gemm.txt
  TSG_HIP_LIB=$PWD/tools/_ablate/$v.so python tools/gemm_f32s_time.py 2>&1 | grep -v amdgpu | cut -c1-90 >> gpurun_out/r4q/gemm.txt
done
for v in base wgrad_noslp; do
  echo "== $v" >> gpurun_out/r4q/gemm.txt
  if [ $v = base ]; then python tools/wgrad_lstm_probe.py 2>&1 | grep -v amdgpu | cut -c1-100 >> gpurun_out/r4q/gemm.txt
  else TSG_HIP_LIB=$PWD/tools/_ablate/$v.so python tools/wgrad_lstm_probe.py 2>&1 | grep -v amdgpu | cut -c1-100 >> gpurun_out/r4q/gemm.txt; fi
done
done
cat gpurun_out/r4q/gemm.txt
