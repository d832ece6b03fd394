#!/bin/bash
# LSTM backward: 2-slot ring (TSG_LSTM_RING=2) vs 4 slots -- tests, step time, traffic
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r4bt; rm -rf $O; mkdir -p $O
(TSG_LSTM_RING=2 timeout 900 python -m pytest tests/test_lstm_gpu.py -x -q -m gpu 2>&1 | tail -3) > $O/pytest_ring2.txt
for shape in "128 128 512" "128 256 512" "64 128 512" "32 512 512" "128 128 256"; do for dt in 2 1; do for r in 4 2 4 2; do
  echo "== $shape dtype $dt RING=$r" >> $O/ring.txt
  TSG_LSTM_RING=$r TSG_REC_DTYPE=$dt TSG_BM=1 timeout 300 python tools/lstm_bench.py $shape 2>&1 | grep "persistent backward" | cut -c1-100 >> $O/ring.txt
done; done; done
for r in 4 2; do
  TSG_LSTM_RING=$r timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pf$r -o p -- python3 tools/lstm_bench.py 128 128 512 > /dev/null 2>&1
  TSG_LSTM_RING=$r timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pw$r -o p -- python3 tools/lstm_bench.py 128 128 512 > /dev/null 2>&1
  A=$(find $O/pf$r -name "*counter_collection.csv" | head -1); B=$(find $O/pw$r -name "*counter_collection.csv" | head -1)
  echo "== RING=$r (fp32 recurrence, [128,128,512])" >> $O/ring_traffic.txt
  python3 tools/pmc_traffic_by_kernel.py $A $B 6 | grep "lstm_\|^#" | cut -c1-160 >> $O/ring_traffic.txt
done
find $O -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
cat $O/pytest_ring2.txt; paste - - < $O/ring.txt | cut -c1-130; cat $O/ring_traffic.txt
