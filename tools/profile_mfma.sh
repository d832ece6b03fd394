#!/bin/bash
# MFMA utilisation per kernel of the bench step (PMC pass, no other trace domains).  Through gpurun.
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
O=$PWD/gpurun_out/prof_mfma; rm -rf $O; mkdir -p $O
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $O/p -o p -- python3 bench.py --steps 6 --warmup 3 --cpu-sample 0 --no-alt --no-micro > $O/bench.json 2> $O/bench.err
C=$(find $O/p -name "*counter_collection.csv" | head -1)
python3 tools/pmc_mfma_util.py $C 16 > $O/mfma_util.txt 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $O/k -o p -- python3 tools/k2_split_only.py 6 64 128 128 1024 8 > /dev/null 2>&1
C=$(find $O/k -name "*counter_collection.csv" | head -1)
echo "== K2 split kernels [64,128,128,1024,h8]" >> $O/mfma_util.txt
python3 tools/pmc_mfma_util.py $C 6 >> $O/mfma_util.txt 2>&1
find $O -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
cat $O/mfma_util.txt | cut -c1-200
