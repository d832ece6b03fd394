#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=$PWD/gpurun_out/prof_k2; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python3 tools/k2_bwd_time.py > $O/time.txt 2>&1
T=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python3 tools/trace_medians.py $T 12 > $O/medians.txt
cat $O/time.txt | tail -5; cat $O/medians.txt | cut -c1-180
find $O -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
