#!/bin/bash
# GPU job of round 4 (bh): where the remaining small torch launches of the f32s / bf16 step come from (by autograd node / call site)
O=gpurun_out/r4bh; rm -rf $O; mkdir -p $O
MODE=f32s timeout 600 python tools/glue_sites.py > $O/glue_f32s.txt 2>&1
MODE=bf16 timeout 600 python tools/glue_sites.py > $O/glue_bf16.txt 2>&1
head -70 $O/glue_f32s.txt | cut -c1-220
