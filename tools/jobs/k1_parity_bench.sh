#!/bin/bash
# K1 parity suites, then the default bench (in-step K1g time).  usage: k1_parity_bench.sh OUT
O=gpurun_out/$1; mkdir -p $O
(timeout 2400 python -m pytest tests/test_scdm_gpu.py tests/test_bf16_storage_gpu.py tests/test_config1_gpu.py tests/test_fullsize_gpu.py tests/test_config_anet256_gpu.py -q -m gpu -x 2>&1 | tail -15) > $O/pytest.txt
cat $O/pytest.txt
python bench.py > $O/bench.json 2> $O/bench.log; tail -3 $O/bench.log
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","roofline")})
print([ (k["name"],k.get("us"),k.get("frac")) for k in d.get("kernels",[])][:12] if isinstance(d.get("kernels"),list) else d.get("kernels"))
PY
