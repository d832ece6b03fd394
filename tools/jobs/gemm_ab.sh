#!/bin/bash
# f32s GEMM: parity suites, then the tree's library vs a variant, stand-alone shapes and the step.   usage: gemm_ab.sh OUT variant
O=gpurun_out/$1; mkdir -p $O
(timeout 2400 python -m pytest tests/test_gemm_f32s_gpu.py tests/test_head_gemm_gpu.py tests/test_grad_sink_gpu.py tests/test_lstm_gpu.py tests/test_fullsize_gpu.py -q -m gpu 2>&1 | tail -3) | tee $O/pytest.txt
for rep in 1 2; do
  for v in tree $2; do
    if [ $v = tree ]; then unset TSG_HIP_LIB; else export TSG_HIP_LIB=tools/_ablate/$v.so; fi
    echo "== $v" >> $O/gemm.txt
    python tools/gemm_f32s_time.py 2>/dev/null | cut -c1-75 >> $O/gemm.txt
    python bench.py --steps 20 --warmup 5 --no-alt --no-micro --cpu-sample 0 --graph on 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); g=d.get('graph_replay_in_process') or {}
print('step eager', d['eager']['ms_per_step'], 'graph', g.get('ms_per_step'), 'ms  K1g', d['roofline']['mean_launch_us'])" >> $O/gemm.txt
  done
done
cat $O/gemm.txt
