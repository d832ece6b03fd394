#!/bin/bash
mkdir -p gpurun_out/r4au
for T in 32 64 128 256; do TSG_REC_DTYPE=2 TSG_BM=1 python tools/lstm_bench.py 128 $T 512 2>&1 | grep -v amdgpu | tail -2 | cut -c1-160; done > gpurun_out/r4au/lstm_T.txt
cat gpurun_out/r4au/lstm_T.txt
