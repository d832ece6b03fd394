#!/bin/bash
mkdir -p gpurun_out/r4av
(timeout 3000 python -m pytest tests -x -q -m gpu --durations=8 2>&1 | tail -16) > gpurun_out/r4av/durations.txt
cat gpurun_out/r4av/durations.txt
