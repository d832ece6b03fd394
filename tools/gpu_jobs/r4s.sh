#!/bin/bash
mkdir -p gpurun_out/r4s
python tools/cat_sites.py > gpurun_out/r4s/cats.txt 2>&1
head -60 gpurun_out/r4s/cats.txt
