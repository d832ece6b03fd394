#!/bin/bash
# SQ counters + kernel trace of the wgrad kernel (run through gpurun).  Output: gpurun_out/prof_wgrad/summary.txt
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=$PWD/gpurun_out/prof_wgrad; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python3 tools/wgrad_only.py 20 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $O/sq1 -o p -- python3 tools/wgrad_only.py 6 > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $O/sq2 -o p -- python3 tools/wgrad_only.py 6 > /dev/null 2>&1
rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_WAIT_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_WAVES SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL --kernel-trace --output-format csv -d $O/sq3 -o p -- python3 tools/wgrad_only.py 6 > /dev/null 2>&1
for d in sq1 sq2 sq3; do
  C=$(find $O/$d -name "*counter_collection.csv" | head -1)
  echo "== $d" >> $O/summary.txt
  python3 tools/pmc_summary.py $C wgrad_split >> $O/summary.txt 2>&1
done
T=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python3 tools/trace_medians.py $T 10 | grep -i "wgrad\|name" >> $O/summary.txt
cat $O/summary.txt
find $O -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
