#!/bin/bash
# default bench with and without the runtime's graph packet capture.  usage: bench_env.sh OUT
O=gpurun_out/$1; mkdir -p $O
for e in "X=1" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0"; do
  env $e python bench.py --cpu-sample 0 --no-micro > $O/b.json 2> $O/b.log
  python - "$e" $O/b.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
g = d.get("graph_replay_in_process") or {}
print(sys.argv[1], "| value", d["value"], d["value_mode"][:12], "| eager", d["eager"]["ms_per_step"], "| graph", g.get("ms_per_step"), "host enqueue", g.get("host_enqueue_ms_per_step_by_rank"),
      "| bf16 graph", (d.get("graph_replay_bf16") or {}).get("ms_per_step"), (d.get("graph_replay_bf16") or {}).get("host_enqueue_ms_per_step"), "| K1g", d["roofline"]["mean_launch_us"])
PY
done
