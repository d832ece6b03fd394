#!/bin/bash
# soak of the persistent LSTM kernels incl. the padded grids, then long bench runs at the per-GPU shard shapes (error sinks checked by bench)
O=gpurun_out/r4bq; rm -rf $O; mkdir -p $O
timeout 1500 python tools/lstm_soak.py 300 > $O/lstm_soak.txt 2>&1
C="--cpu-sample 0 --no-alt --no-micro"
timeout 600 python bench.py --B 16 --T 512 --N 25 --steps 400 --warmup 5 $C > $O/bench_shard_c4_400steps.json 2> $O/c4.err
timeout 600 python bench.py --B 16 --T 256 --N 25 --dtype bf16 --steps 400 --warmup 5 $C > $O/bench_shard_c3_bf16_400steps.json 2> $O/c3.err
grep -v amdgpu $O/lstm_soak.txt | tail -16; cut -c1-260 $O/bench_shard_c4_400steps.json; cut -c1-260 $O/bench_shard_c3_bf16_400steps.json
