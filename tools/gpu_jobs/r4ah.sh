#!/bin/bash
mkdir -p gpurun_out/r4ah
python tools/wgrad_bf16_time.py 2>&1 | grep -v amdgpu > gpurun_out/r4ah/wgrad_bf16.txt
cat gpurun_out/r4ah/wgrad_bf16.txt
