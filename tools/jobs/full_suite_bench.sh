#!/bin/bash
# The whole GPU suite (no -x: every failure is listed), then the default bench, then the other BASELINE configs.   usage: full_suite_bench.sh OUT [configs: 0/1]
O=gpurun_out/$1; mkdir -p $O
(timeout 3300 python -m pytest tests -q -m gpu 2>&1 | tail -25) > $O/pytest_gpu.txt
cat $O/pytest_gpu.txt
python bench.py > $O/bench_default.json 2> $O/bench_default.log; tail -4 $O/bench_default.log
python - <<PY
import json
d=json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print({k:d.get(k) for k in ("value","ms_per_step","value_mode","value_f32","value_bf16","skipped_updates")})
r=d["roofline"]; print({k:r.get(k) for k in ("frac","mean_launch_us","around_call_mean_us","launches_timed")})
print("eager",d["eager"],"graph",d.get("graph_replay_in_process"),"bf16 graph",d.get("graph_replay_bf16"))
PY
if [ "${2:-0}" = "1" ]; then bash tools/bench_configs.sh $O/configs 10; fi
