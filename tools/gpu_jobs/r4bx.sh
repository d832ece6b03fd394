#!/bin/bash
# LSTM backward with the non-temporal hint (product build) vs without (-DTSG_LSTM_NO_NT): tests on the product build, train step A/B
O=$PWD/gpurun_out/r4bx; rm -rf $O; mkdir -p $O
NONT=$PWD/tools/_ablate/lstm_nont.so
(timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -3) > $O/pytest_gpu_full.txt
timeout 900 python tools/lstm_soak.py 200 > $O/lstm_soak.txt 2>&1
C="--cpu-sample 0 --no-alt --no-micro --graph on"
for i in 1 2 3; do for v in nont nt; do
  if [ $v = nont ]; then export TSG_HIP_LIB=$NONT; else unset TSG_HIP_LIB; fi
  echo "== $v f32s" >> $O/ab.txt; python bench.py $C 2>/dev/null | cut -c1-330 >> $O/ab.txt
  echo "== $v bf16" >> $O/ab.txt; python bench.py --dtype bf16 $C 2>/dev/null | cut -c1-330 >> $O/ab.txt
done; done
unset TSG_HIP_LIB
cat $O/pytest_gpu_full.txt; grep -v amdgpu $O/lstm_soak.txt | tail -2; grep -o "==.*\|\"ms_per_step\": [0-9.]*" $O/ab.txt | paste - -
