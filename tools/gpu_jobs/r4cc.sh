#!/bin/bash
# L2 hit / miss counts of the persistent LSTM kernels (PMC)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r4cc; rm -rf $O; mkdir -p $O
for dt in 2 1; do
  TSG_REC_DTYPE=$dt TSG_BM=1 timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $O/p$dt -o p -- python3 tools/lstm_bench.py 128 128 512 > /dev/null 2>&1
  C=$(find $O/p$dt -name "*counter_collection.csv" | head -1)
  echo "== dtype $dt" >> $O/l2.txt
  python3 tools/pmc_summary.py $C lstm_fwd_persist >> $O/l2.txt 2>&1
  echo "-- bwd" >> $O/l2.txt
  python3 tools/pmc_summary.py $C lstm_bwd_persist2 >> $O/l2.txt 2>&1
done
find $O -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
cat $O/l2.txt
