#!/bin/bash
# round 5, session 23: the train step with the backward's operand DMAs: non-temporal never / fp32 storage only (default) / always, vs the round-4 loads
O=gpurun_out/r5w; mkdir -p $O
for rep in 1 2 3; do
for lib in shufflingvideosfortsg_amd/libtsg_hip.so tools/_ablate/bnt0.so tools/_ablate/bnt2.so tools/_ablate/prevlstm.so; do
  for dtype in f32s bf16; do
      echo "lib=$lib $dtype: $(TSG_HIP_LIB=$lib python bench.py --dtype $dtype --no-alt --cpu-sample 0 --no-micro --graph on 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])')" >> $O/bench.txt
  done
done
done
sort $O/bench.txt
