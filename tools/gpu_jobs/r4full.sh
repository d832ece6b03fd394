#!/bin/bash
# GPU job of round 4: the whole GPU suite, smoke(), the default bench line (what the driver runs), the bf16-storage line
mkdir -p gpurun_out/r4full
(timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -5) > gpurun_out/r4full/pytest_gpu_full.txt
(python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2) > gpurun_out/r4full/smoke.txt
python bench.py > gpurun_out/r4full/bench_default.json 2> gpurun_out/r4full/bench_default.err
python bench.py --dtype bf16 --no-alt --cpu-sample 0 > gpurun_out/r4full/bench_bf16.json 2> gpurun_out/r4full/bench_bf16.err
cat gpurun_out/r4full/pytest_gpu_full.txt gpurun_out/r4full/smoke.txt; cut -c1-400 gpurun_out/r4full/bench_default.json; cut -c1-300 gpurun_out/r4full/bench_bf16.json
