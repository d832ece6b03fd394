#!/bin/bash
# GPU job of round 4 (e): the full GPU suite, smoke(), the default bench line.
mkdir -p gpurun_out/r4e
(timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -15) > gpurun_out/r4e/pytest_gpu_full.txt
(python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -3) > gpurun_out/r4e/smoke.txt
(python bench.py 2>gpurun_out/r4e/bench_err.txt | tail -1) > gpurun_out/r4e/bench_default.json
cat gpurun_out/r4e/pytest_gpu_full.txt gpurun_out/r4e/smoke.txt; cut -c1-400 gpurun_out/r4e/bench_default.json
