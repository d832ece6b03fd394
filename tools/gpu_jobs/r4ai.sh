#!/bin/bash
mkdir -p gpurun_out/r4ai
for i in 1 2 3 4 5 6 7 8; do
(timeout 1200 python -m pytest tests/test_bench_gpu.py -x -q -m gpu -k "multi_gpu or contract" 2>&1 | tail -40) > gpurun_out/r4ai/pytest_$i.txt
tail -1 gpurun_out/r4ai/pytest_$i.txt
done
(timeout 600 python -m pytest tests/test_dp_rccl_gpu.py -x -q -m gpu 2>&1 | tail -2)
