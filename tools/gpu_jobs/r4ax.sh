#!/bin/bash
mkdir -p gpurun_out/r4ax
( time python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) 2>&1 | tail -5 > gpurun_out/r4ax/smoke.txt
python bench.py > gpurun_out/r4ax/bench_default.json 2> gpurun_out/r4ax/err.txt
cat gpurun_out/r4ax/smoke.txt; cut -c1-200 gpurun_out/r4ax/bench_default.json; python -c "
import json; d=json.loads(open('gpurun_out/r4ax/bench_default.json').read().strip().splitlines()[-1]); print(d['cpu_baseline'], d['roofline']['frac'])"
