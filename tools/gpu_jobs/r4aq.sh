#!/bin/bash
mkdir -p gpurun_out/r4aq
python bench.py --no-alt --cpu-sample 0 > gpurun_out/r4aq/bench.json 2> gpurun_out/r4aq/err.txt; echo rc=$?
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4aq/bench.json').read().strip().splitlines()[-1])
for n,r in d['kernels'].items():
    if 'gemm' in n or 'wgrad' in n: print(n, r.get('mean_us'), r.get('achieved_TFLOPs'), r.get('mfma_frac'))
PY
tail -3 gpurun_out/r4aq/err.txt
