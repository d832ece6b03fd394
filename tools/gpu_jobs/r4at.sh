#!/bin/bash
mkdir -p gpurun_out/r4at
(timeout 1800 python -m pytest tests/test_bench_gpu.py tests/test_match_head_gpu.py -x -q -m gpu 2>&1 | tail -2) > gpurun_out/r4at/pytest.txt
python bench.py --dtype bf16 --graph on --cpu-sample 0 --no-alt > gpurun_out/r4at/bench_bf16.json 2>/dev/null
cat gpurun_out/r4at/pytest.txt; cut -c1-300 gpurun_out/r4at/bench_bf16.json
