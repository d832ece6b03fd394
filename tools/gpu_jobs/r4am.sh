#!/bin/bash
mkdir -p gpurun_out/r4am
for cfg in 0 1 2 3; do
echo "== TSG_WGRAD_TR_CFG=$cfg (0: 1 sub-chunk, ring 3; 1: 1, ring 4; 2: 2 sub-chunks, ring 3; 3: 2, ring 2)"
(TSG_WGRAD_TR_CFG=$cfg timeout 900 python -m pytest tests/test_wgrad_gpu.py -x -q -m gpu -k "bf16" 2>&1 | tail -1)
TSG_WGRAD_TR_CFG=$cfg python tools/wgrad_bf16_time.py 2>&1 | grep -v amdgpu | cut -c1-100
done > gpurun_out/r4am/time.txt
cat gpurun_out/r4am/time.txt
