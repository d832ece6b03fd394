#!/bin/bash
# Profiling session of the bench step on the GPU box (run through gpurun): rocprofv3 kernel-trace statistics of `bench.py` in the f32s headline mode
# and in the bf16 storage mode -> per-kernel medians, per-step breakdown, the --stats summary; then the PMC traffic pass of the roofline kernel
# (separate --pmc passes, no trace domains beside them).  Summaries under gpurun_out/<OUT>/ -- copy the ones to be judged to profiles/rN/.
#     usage: tools/profile_step.sh [OUT]
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=$PWD/gpurun_out/${1:-prof}; rm -rf $O; mkdir -p $O
for mode in f32s bf16; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$mode -o bench -- python3 bench.py --dtype $mode --steps 30 --warmup 5 --cpu-sample 0 --no-alt --no-micro --graph off > $O/bench_trace_$mode.json 2> $O/bench_trace_$mode.err
  T=$(find $O/trace_$mode -name "*kernel_trace.csv" | head -1)
  python3 tools/trace_medians.py $T 70 > $O/bench_gmd_kernel_medians_$mode.txt
  python3 tools/step_breakdown.py $T --glue > $O/bench_gmd_step_breakdown_$mode.txt 2>&1
  S=$(find $O/trace_$mode -name "*kernel_stats.csv" | head -1); [ -n "$S" ] && head -40 $S > $O/bench_gmd_kernel_stats_${mode}_summary.csv
  rm -rf $O/trace_$mode
done
python3 tools/profile_k1_traffic.py $O/k1_pmc_traffic.json > $O/k1_pmc_traffic.log 2>&1
head -12 $O/bench_gmd_step_breakdown_f32s.txt; head -10 $O/bench_gmd_step_breakdown_bf16.txt; head -14 $O/bench_gmd_kernel_medians_f32s.txt | cut -c1-200; cat $O/k1_pmc_traffic.log | tail -2
python3 - $O <<'PY'
import json, sys
for m in ("f32s", "bf16"):
    try:
        d = json.loads(open(f"{sys.argv[1]}/bench_trace_{m}.json").read().strip().splitlines()[-1])
        r = d["roofline"]
        print(m, "profiled run:", d["ms_per_step"], "ms/step; K1g fwd by its own events", r["mean_launch_us"], "us, around the call", r.get("around_call_mean_us"))
    except Exception as e:  # noqa: BLE001
        print(m, "no bench line:", e)
PY
