#!/bin/bash
mkdir -p gpurun_out/r4w
python tools/gemm_shapes.py 2>/dev/null > gpurun_out/r4w/gemm_shapes.txt
cat gpurun_out/r4w/gemm_shapes.txt | head -50
