#!/bin/bash
# Round-2 late profiling session (after the weight-gradient kernel): kernel-trace medians + step breakdown of the bench step, bench JSONs.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=$PWD/gpurun_out/prof_r2b; rm -rf $O; mkdir -p $O
python3 bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
python3 bench.py --predictor self_attn --steps 10 --warmup 3 --no-alt --cpu-sample 0 > $O/bench_self_attn.json 2> $O/bench_self_attn.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python3 bench.py --steps 30 --warmup 5 --cpu-sample 0 --no-alt --no-micro > $O/bench_trace.json 2> $O/bench_trace.err
T=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python3 tools/trace_medians.py $T 60 > $O/bench_kernel_medians.txt
python3 tools/step_breakdown.py $T > $O/bench_step_breakdown.txt 2>&1
python3 tools/gemm_shapes.py > $O/gemm_shapes.txt 2>&1
find $O -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
tail -1 $O/bench_default.json | cut -c1-400; tail -1 $O/bench_self_attn.json | cut -c1-300; head -12 $O/bench_step_breakdown.txt; head -25 $O/gemm_shapes.txt
