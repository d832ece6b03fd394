#!/bin/bash
mkdir -p gpurun_out/r4af
(python bench.py --predictor self_attn --steps 8 --warmup 3 --no-alt --cpu-sample 0 --no-micro 2>&1 | tail -1 | cut -c1-200) > gpurun_out/r4af/variants.txt
(python bench.py --fwd-only --B 32 --T 64 --d 512 --dtype f32 --steps 8 --warmup 3 --no-alt --cpu-sample 0 --no-micro 2>&1 | tail -1 | cut -c1-200) >> gpurun_out/r4af/variants.txt
(python bench.py --B 64 --T 256 --N 25 --steps 5 --warmup 2 --no-alt --cpu-sample 0 --no-micro 2>&1 | tail -1 | cut -c1-200) >> gpurun_out/r4af/variants.txt
(python bench.py --B 16 --T 512 --N 25 --dtype bf16 --steps 5 --warmup 2 --no-alt --cpu-sample 0 --no-micro 2>&1 | tail -1 | cut -c1-200) >> gpurun_out/r4af/variants.txt
(python bench.py --B 16 --T 512 --N 25 --steps 5 --warmup 2 --no-alt --cpu-sample 0 --no-micro 2>&1 | tail -1 | cut -c1-200) >> gpurun_out/r4af/variants.txt
(python bench.py --dtype f32 --steps 5 --warmup 2 --no-alt --cpu-sample 0 --no-micro 2>&1 | tail -1 | cut -c1-200) >> gpurun_out/r4af/variants.txt
cat gpurun_out/r4af/variants.txt
