#!/bin/bash
# LSTM forward: non-temporal stores of R / Cs only (-DTSG_LSTM_FWD_NT_ST) vs the product build
O=$PWD/gpurun_out/r4by; rm -rf $O; mkdir -p $O
V=$PWD/tools/_ablate/lstm_fwdst.so
for shape in "128 128 512" "128 256 512" "64 20 512" "32 512 512"; do for dt in 2 1; do for v in base st base st; do
  echo "== $shape dtype $dt $v" >> $O/st.txt
  if [ $v = st ]; then export TSG_HIP_LIB=$V; else unset TSG_HIP_LIB; fi
  TSG_REC_DTYPE=$dt TSG_BM=1 timeout 300 python tools/lstm_bench.py $shape 2>&1 | grep "rec dtype" | cut -c40-110 >> $O/st.txt
done; done; done
paste - - < $O/st.txt | cut -c1-120
