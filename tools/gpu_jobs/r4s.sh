#!/bin/bash
mkdir -p gpurun_out/r4s
python tools/glue_sites.py > gpurun_out/r4s/glue.txt 2>&1
python tools/cat_sites.py > gpurun_out/r4s/cats.txt 2>&1
head -75 gpurun_out/r4s/glue.txt; cat gpurun_out/r4s/cats.txt
