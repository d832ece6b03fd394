#!/bin/bash
# new bf16 weight-gradient tests (layouts, ring variants)
mkdir -p gpurun_out/r4ba
timeout 900 python -m pytest tests/test_wgrad_gpu.py -x -q -m gpu > gpurun_out/r4ba/pytest_wgrad.txt 2>&1
tail -15 gpurun_out/r4ba/pytest_wgrad.txt
