#!/bin/bash
mkdir -p gpurun_out/r4aw
python tools/oracle_threads_probe.py > gpurun_out/r4aw/threads.txt 2>&1
cat gpurun_out/r4aw/threads.txt
