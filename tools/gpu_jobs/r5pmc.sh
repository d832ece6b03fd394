#!/bin/bash
# GPU job of round 5: memory-side traffic per kernel of the final step, f32s and bf16 storage (PMC: FETCH_SIZE and WRITE_SIZE in separate passes),
# and the K1g forward's traffic JSON that bench.py's roofline cites (tools/profile_r4_k1_traffic.py: same kernel, round-5 tree)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r5pmc; rm -rf $O; mkdir -p $O
for mode in f32s bf16; do
  C="--dtype $mode --steps 3 --warmup 2 --cpu-sample 0 --no-alt --no-micro --graph off"
  timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pf$mode -o p -- python3 bench.py $C > $O/pf.json 2> $O/pf.err
  timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pw$mode -o p -- python3 bench.py $C > $O/pw.json 2> $O/pw.err
  A=$(find $O/pf$mode -name "*counter_collection.csv" | head -1); B=$(find $O/pw$mode -name "*counter_collection.csv" | head -1)
  python3 tools/pmc_traffic_by_kernel.py $A $B 22 > $O/step_traffic_by_kernel_$mode.txt 2>&1
done
find $O -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
timeout 900 python3 tools/profile_r4_k1_traffic.py $O/k1_pmc_traffic.json > $O/k1_traffic.log 2>&1
cut -c1-150 $O/step_traffic_by_kernel_f32s.txt | head -10; cut -c1-150 $O/step_traffic_by_kernel_bf16.txt | head -8; cat $O/k1_pmc_traffic.json | head -30
