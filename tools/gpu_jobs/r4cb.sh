#!/bin/bash
# GPU job of round 4 (cb): memory-side traffic per kernel of the final step, f32s and bf16 storage (PMC: FETCH_SIZE and WRITE_SIZE in separate passes)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r4cb; rm -rf $O; mkdir -p $O
for mode in f32s bf16; do
  C="--dtype $mode --steps 3 --warmup 2 --cpu-sample 0 --no-alt --no-micro --graph off"
  timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pf$mode -o p -- python3 bench.py $C > $O/pf.json 2> $O/pf.err
  timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pw$mode -o p -- python3 bench.py $C > $O/pw.json 2> $O/pw.err
  A=$(find $O/pf$mode -name "*counter_collection.csv" | head -1); B=$(find $O/pw$mode -name "*counter_collection.csv" | head -1)
  python3 tools/pmc_traffic_by_kernel.py $A $B 22 > $O/step_traffic_by_kernel_$mode.txt 2>&1
done
find $O -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
cut -c1-150 $O/step_traffic_by_kernel_f32s.txt | head -8; cut -c1-150 $O/step_traffic_by_kernel_bf16.txt | head -14
