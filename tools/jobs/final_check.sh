#!/bin/bash
# What the driver runs at round end, in its order, on one box: the GPU suite with -x, smoke(), the default bench; then the other configs and the
# self-attention-predictor variant.   usage: final_check.sh OUT
O=gpurun_out/$1; mkdir -p $O
(timeout 3300 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -8) > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt
(python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2) > $O/smoke.txt; cat $O/smoke.txt
python bench.py > $O/bench_default.json 2> $O/bench_default.log; tail -3 $O/bench_default.log
python - <<PY
import json
d=json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print({k:d.get(k) for k in ("value","ms_per_step","value_mode","value_f32","value_bf16","skipped_updates")})
r=d["roofline"]; print({k:r.get(k) for k in ("frac","mean_launch_us","around_call_mean_us","launches_timed","traffic")})
for k,v in d["kernels"].items():
    if "mha" in k or "scdm" in k: print(k, v.get("mean_us"), v.get("frac"), v.get("cached_mean_us"))
PY
bash tools/bench_configs.sh $O/configs 10
python bench.py --predictor self_attn --cpu-sample 0 --no-micro > $O/bench_self_attn.json 2> $O/bench_self_attn.log; tail -2 $O/bench_self_attn.log
python - <<PY
import json
d=json.loads(open("$O/bench_self_attn.json").read().strip().splitlines()[-1])
print("self_attn predictor:", {k:d.get(k) for k in ("value","ms_per_step","value_mode","value_f32","value_bf16")})
for k,v in d["kernels"].items():
    if "mha" in k: print(k, v.get("mean_us"), v.get("frac"))
PY
