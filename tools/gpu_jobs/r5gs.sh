#!/bin/bash
O=gpurun_out/r5gs; mkdir -p $O
python tools/gemm_shapes.py > $O/shapes.txt 2>&1; grep -v "amdgpu\|Warning\|warn" $O/shapes.txt | head -45 | cut -c1-220
