#!/bin/bash
# round 5, session 37: M tile of the f32s GEMM at the one-round shapes (N = 1024: 256 tiles of 256 rows vs 512 of 128)
O=gpurun_out/r5tm; mkdir -p $O
for rep in 1 2; do
for tm in 0 128; do
  echo "== TSG_GEMM_TM=$tm" >> $O/gemm.txt
  TSG_GEMM_TM=$tm python tools/gemm_f32s_time.py 2>/dev/null | grep "^\[" | cut -c1-90 >> $O/gemm.txt
done
done
cat $O/gemm.txt
