#!/bin/bash
# f32s GEMM / weight-gradient kernels: the tree's library vs a variant, stand-alone and in the step.   usage: gemm_young_ab.sh OUT variant
O=gpurun_out/$1; mkdir -p $O
for rep in 1 2; do
  for v in tree $2; do
    if [ $v = tree ]; then unset TSG_HIP_LIB; else export TSG_HIP_LIB=tools/_ablate/$v.so; fi
    echo "== $v" >> $O/gemm.txt
    python tools/gemm_f32s_time.py 2>/dev/null | cut -c1-75 >> $O/gemm.txt
    python tools/wgrad_time.py 2>/dev/null | head -6 | cut -c1-120 >> $O/gemm.txt
    python bench.py --steps 20 --warmup 5 --no-alt --no-micro --cpu-sample 0 --graph off 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('step', d['ms_per_step'], 'ms  K1g', d['roofline']['mean_launch_us'])" >> $O/gemm.txt
  done
done
cat $O/gemm.txt
