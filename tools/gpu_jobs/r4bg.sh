#!/bin/bash
# GPU job of round 4 (bg): MFMA utilisation of the train step per kernel (PMC passes; f32s and bf16 storage)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r4bg; rm -rf $O; mkdir -p $O
C="--steps 4 --warmup 2 --cpu-sample 0 --no-alt --no-micro --graph off"
for mode in f32s bf16; do
  timeout 900 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $O/p1_$mode -o p -- python3 bench.py --dtype $mode $C > $O/p1_$mode.json 2> $O/p1_$mode.err
  timeout 900 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace --output-format csv -d $O/p2_$mode -o p -- python3 bench.py --dtype $mode $C > $O/p2_$mode.json 2> $O/p2_$mode.err
  A=$(find $O/p1_$mode -name "*counter_collection.csv" | head -1); B=$(find $O/p2_$mode -name "*counter_collection.csv" | head -1)
  (cat $A; tail -n +2 $B) > $O/both_$mode.csv
  python3 tools/pmc_mfma_util.py $O/both_$mode.csv 24 > $O/mfma_utilisation_$mode.txt 2>&1
  rm -f $O/both_$mode.csv
done
find $O -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
cut -c1-200 $O/mfma_utilisation_f32s.txt; tail -3 $O/p1_f32s.err
