#!/bin/bash
# Round-5 profiling session on the GPU box (run through gpurun).  Summaries under gpurun_out/prof_r5/ (copied to profiles/r5/).
#  kernel-trace statistics of the bench step (f32s headline mode, bf16 storage mode): per-kernel medians + per-step breakdown + foreign launches
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=$PWD/gpurun_out/prof_r5; rm -rf $O; mkdir -p $O
for mode in f32s bf16; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$mode -o bench -- python3 bench.py --dtype $mode --steps 30 --warmup 5 --cpu-sample 0 --no-alt --no-micro > $O/bench_trace_$mode.json 2> $O/bench_trace_$mode.err
  T=$(find $O/trace_$mode -name "*kernel_trace.csv" | head -1)
  python3 tools/trace_medians.py $T 70 > $O/bench_gmd_kernel_medians_$mode.txt
  python3 tools/step_breakdown.py $T --glue > $O/bench_gmd_step_breakdown_$mode.txt 2>&1
  S=$(find $O/trace_$mode -name "*kernel_stats.csv" | head -1); [ -n "$S" ] && head -40 $S > $O/bench_gmd_kernel_stats_${mode}_summary.csv
  rm -rf $O/trace_$mode
done
head -12 $O/bench_gmd_step_breakdown_f32s.txt; head -10 $O/bench_gmd_step_breakdown_bf16.txt; head -14 $O/bench_gmd_kernel_medians_f32s.txt | cut -c1-200
