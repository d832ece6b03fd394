#!/bin/bash
# GPU job of round 4 (bf): kernel trace of config 3's single-GPU batch (B=64, T=256, N=25), f32s and bf16
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r4bf; rm -rf $O; mkdir -p $O
C="--cpu-sample 0 --no-alt --no-micro"
for mode in f32s bf16; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$mode -o bench -- python3 bench.py --B 64 --T 256 --N 25 --dtype $mode --steps 20 --warmup 5 $C > $O/bench_trace_$mode.json 2> $O/bench_trace_$mode.err
  T=$(find $O/trace_$mode -name "*kernel_trace.csv" | head -1)
  python3 tools/trace_medians.py $T 40 > $O/bench_config3_kernel_medians_$mode.txt
  python3 tools/step_breakdown.py $T > $O/bench_config3_step_breakdown_$mode.txt 2>&1
done
find $O -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
head -12 $O/bench_config3_step_breakdown_f32s.txt; head -24 $O/bench_config3_kernel_medians_f32s.txt | cut -c1-60,120-200
