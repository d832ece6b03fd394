#!/bin/bash
# Round-3 profiling session after the role-specialised K1 forward: kernel-trace medians + step breakdown (with the glue and GEMM
# lists) of the bench step in both modes.  Summaries under gpurun_out/prof_r3c/, copied to profiles/r3/ as *_v3_*.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=$PWD/gpurun_out/prof_r3c; rm -rf $O; mkdir -p $O
for mode in f32s bf16; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$mode -o bench -- python3 bench.py --dtype $mode --steps 30 --warmup 5 --cpu-sample 0 --no-alt --no-micro --graph off > $O/bench_trace_$mode.json 2> $O/bench_trace_$mode.err
  T=$(find $O/trace_$mode -name "*kernel_trace.csv" | head -1)
  python3 tools/trace_medians.py $T 70 > $O/bench_gmd_kernel_medians_$mode.txt
  python3 tools/step_breakdown.py $T --glue --gemms > $O/bench_gmd_step_breakdown_$mode.txt 2>&1
  S=$(find $O/trace_$mode -name "*kernel_stats.csv" | head -1); [ -n "$S" ] && head -40 $S > $O/bench_gmd_kernel_stats_$mode.csv
done
find $O -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
head -12 $O/bench_gmd_step_breakdown_*.txt
