#!/bin/bash
mkdir -p gpurun_out/r4ae
(timeout 900 python -m pytest tests/test_wgrad_gpu.py -x -q -m gpu 2>&1 | tail -3) > gpurun_out/r4ae/pytest.txt
cat gpurun_out/r4ae/pytest.txt
