#!/bin/bash
# round 5: long runs of the final tree (400 steps per mode, eager + graph replay): finite losses, clean error words
O=gpurun_out/r5soak; mkdir -p $O
for dtype in f32s bf16; do
  timeout 600 python bench.py --dtype $dtype --steps 400 --warmup 5 --no-alt --cpu-sample 0 --no-micro --graph on 2> $O/soak_$dtype.err | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["dtype"], "steps", d["steps"], "ms", d["ms_per_step"], "eager", d["eager"]["ms_per_step"], "finite", (d.get("graph_replay_in_process") or {}).get("finite"))' >> $O/soak.txt
  tail -2 $O/soak_$dtype.err >> $O/soak.txt
done
timeout 900 python -m pytest tests/test_lstm_soak_gpu.py tests/test_dp_rccl_gpu.py -q -m gpu 2>&1 | tail -1 >> $O/soak.txt
cat $O/soak.txt
