#!/bin/bash
# step breakdown of the bench step in one GEMM mode: tools/profile_mode.sh bf16   (through gpurun)
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
M=${1:-bf16}; O=$PWD/gpurun_out/prof_$M; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python3 bench.py --dtype $M --steps 20 --warmup 5 --cpu-sample 0 --no-alt --no-micro > $O/bench.json 2> $O/bench.err
T=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python3 tools/trace_medians.py $T 30 > $O/medians.txt
python3 tools/step_breakdown.py $T > $O/breakdown.txt 2>&1
find $O -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
head -10 $O/breakdown.txt; head -30 $O/medians.txt | cut -c1-170
