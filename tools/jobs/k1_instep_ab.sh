#!/bin/bash
# K1g forward IN THE STEP (bench.py's own-event timing of the roofline kernel), the tree's library vs variant libraries, alternating processes.
#   usage: k1_instep_ab.sh OUT name,name [reps]
O=gpurun_out/$1; mkdir -p $O
one() {  # label, lib or ""
  if [ -n "$2" ]; then export TSG_HIP_LIB=$2; else unset TSG_HIP_LIB; fi
  python bench.py --steps 20 --warmup 5 --no-alt --no-micro --cpu-sample 0 --graph off 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
k=[v for k,v in d['kernels'].items() if k.startswith('tsg_scdm_gate_bwd')]
print('$1', 'K1g fwd', r['mean_launch_us'], 'us  frac', r['frac'], ' bwd', k[0]['mean_us'] if k else None, ' step', d['ms_per_step'], 'ms')" >> $O/instep.txt
}
for rep in $(seq 1 ${3:-3}); do
  one tree ""
  for v in ${2//,/ }; do one $v tools/_ablate/$v.so; done
done
cat $O/instep.txt
