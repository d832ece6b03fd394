#!/bin/bash
mkdir -p gpurun_out/r4ag
for i in 1 2; do
(python bench.py --dtype bf16 --graph on --no-alt --cpu-sample 0 --no-micro 2>gpurun_out/r4ag/err$i.txt | tail -1 | cut -c1-600)
done > gpurun_out/r4ag/bf16_graph.txt
cat gpurun_out/r4ag/bf16_graph.txt; tail -3 gpurun_out/r4ag/err1.txt
