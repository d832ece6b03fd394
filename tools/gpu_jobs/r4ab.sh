#!/bin/bash
mkdir -p gpurun_out/r4ab
for nap in 0 2 4 6 8 12 16; do
echo "== nap $nap (x64 cycles)"
TSG_LSTM_NAP=$nap TSG_HIP_LIB=$PWD/tools/_ablate/lstm_timing.so TSG_REC_DTYPE=2 TSG_BM=1 python tools/lstm_bench.py 128 128 512 2>&1 | grep "phase ticks" | head -1 | cut -c1-200
TSG_LSTM_NAP=$nap TSG_REC_DTYPE=2 TSG_BM=1 python tools/lstm_bench.py 128 128 512 2>&1 | grep -v amdgpu | tail -1
done > gpurun_out/r4ab/lstm.txt
cat gpurun_out/r4ab/lstm.txt
