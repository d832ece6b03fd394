#!/bin/bash
# round 5, session 7: full GPU suite on the pruned tree, then the step's kernel trace (f32s, bf16)
O=gpurun_out/r5g; mkdir -p $O
(timeout 3000 python -m pytest tests -q -m gpu 2>&1 | grep -v "^$" | tail -15) > $O/pytest_gpu_full.txt
cat $O/pytest_gpu_full.txt
bash tools/profile_r5.sh
