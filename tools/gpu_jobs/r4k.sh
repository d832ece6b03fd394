#!/bin/bash
# GPU job of round 4 (k): which of the two copy eliminations pays -- TSG_NO_COPIES = 0 (neither) / out2 / nn / 1 (both), alternating.
mkdir -p gpurun_out/r4k
for i in 1 2 3; do
  for m in 0 out2 nn 1; do
    (TSG_NO_COPIES=$m python bench.py --no-alt --cpu-sample 0 --no-micro --graph on 2>/dev/null | tail -1 | cut -c1-190 | sed "s/^/$m  /")
  done
done > gpurun_out/r4k/bench_no_copies_ab2.txt
cat gpurun_out/r4k/bench_no_copies_ab2.txt
