#!/bin/bash
# LSTM backward: non-temporal hint on the streaming operands (R, Cs, dOut in; dG out) -- -DTSG_LSTM_NT build vs the product build
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r4bv; rm -rf $O; mkdir -p $O
NT=$PWD/tools/_ablate/lstm_nt.so
(TSG_HIP_LIB=$NT timeout 900 python -m pytest tests/test_lstm_gpu.py -x -q -m gpu 2>&1 | tail -2) > $O/pytest_nt.txt
for shape in "128 128 512" "128 256 512" "32 512 512"; do for dt in 2 1; do for v in base nt base nt; do
  echo "== $shape dtype $dt $v" >> $O/nt.txt
  if [ $v = nt ]; then export TSG_HIP_LIB=$NT; else unset TSG_HIP_LIB; fi
  TSG_REC_DTYPE=$dt TSG_BM=1 timeout 300 python tools/lstm_bench.py $shape 2>&1 | grep "persistent backward" | cut -c1-100 >> $O/nt.txt
done; done; done
for v in base nt; do
  if [ $v = nt ]; then export TSG_HIP_LIB=$NT; else unset TSG_HIP_LIB; fi
  timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pf$v -o p -- python3 tools/lstm_bench.py 128 128 512 > /dev/null 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pw$v -o p -- python3 tools/lstm_bench.py 128 128 512 > /dev/null 2>&1
  A=$(find $O/pf$v -name "*counter_collection.csv" | head -1); B=$(find $O/pw$v -name "*counter_collection.csv" | head -1)
  echo "== $v (fp32 recurrence, [128,128,512])" >> $O/nt_traffic.txt
  python3 tools/pmc_traffic_by_kernel.py $A $B 6 | grep "lstm_bwd_persist\|^#" | cut -c1-160 >> $O/nt_traffic.txt
done
find $O -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
cat $O/pytest_nt.txt; paste - - < $O/nt.txt | cut -c1-120; cat $O/nt_traffic.txt
