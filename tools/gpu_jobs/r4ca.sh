#!/bin/bash
# GPU job of round 4 (ca): kernel trace of the bench step after the weight-gradient / GEMM / glue work (f32s and bf16 storage)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r4ca; rm -rf $O; mkdir -p $O
for mode in f32s bf16; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$mode -o bench -- python3 bench.py --dtype $mode --steps 30 --warmup 5 --cpu-sample 0 --no-alt --no-micro > $O/bench_trace_$mode.json 2> $O/bench_trace_$mode.err
  T=$(find $O/trace_$mode -name "*kernel_trace.csv" | head -1)
  python3 tools/trace_medians.py $T 70 > $O/bench_gmd_kernel_medians_$mode.txt
  python3 tools/step_breakdown.py $T > $O/bench_gmd_step_breakdown_$mode.txt 2>&1
done
find $O -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
cat $O/bench_gmd_step_breakdown_f32s.txt | head -12; head -45 $O/bench_gmd_kernel_medians_f32s.txt | cut -c1-170
