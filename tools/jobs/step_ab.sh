#!/bin/bash
# The train step (eager, 20 steps) with the tree's library vs variant libraries, alternating processes.   usage: step_ab.sh OUT dtype name,name [reps]
O=gpurun_out/$1; mkdir -p $O
for rep in $(seq 1 ${4:-3}); do
  for v in tree ${3//,/ }; do
    if [ $v = tree ]; then unset TSG_HIP_LIB; else export TSG_HIP_LIB=tools/_ablate/$v.so; fi
    python bench.py --dtype $2 --steps 20 --warmup 5 --no-alt --no-micro --cpu-sample 0 --graph on 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); g=d.get('graph_replay_in_process') or {}
print('$v', 'eager', d['eager']['ms_per_step'], 'graph', g.get('ms_per_step'), 'K1g', d['roofline']['mean_launch_us'])" >> $O/step.txt
  done
done
cat $O/step.txt
