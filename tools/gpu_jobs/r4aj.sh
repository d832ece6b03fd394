#!/bin/bash
mkdir -p gpurun_out/r4aj
MODE=bf16 python tools/glue_sites.py 2>/dev/null > gpurun_out/r4aj/glue_bf16.txt
head -70 gpurun_out/r4aj/glue_bf16.txt
