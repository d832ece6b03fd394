#!/bin/bash
# K2 split-precision kernels: rocprofv3 kernel-trace medians and HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes).  Through gpurun.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=$PWD/gpurun_out/prof_k2s; rm -rf $O; mkdir -p $O
for shape in "64 128 128 1024 8" "64 128 20 1024 8"; do
  tag=$(echo $shape | tr ' ' '_')
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/t_$tag -o t -- python3 tools/k2_split_only.py 40 $shape > /dev/null 2>&1
  echo "== [$shape] kernel-trace medians (us)" >> $O/summary.txt
  python3 tools/trace_medians.py $(find $O/t_$tag -name "*kernel_trace.csv" | head -1) 8 | grep -i "split\|delta\|from_ds\|# name" | cut -c1-170 >> $O/summary.txt
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/p_${c}_$tag -o p -- python3 tools/k2_split_only.py 4 $shape > /dev/null 2>&1
    C=$(find $O/p_${c}_$tag -name "*counter_collection.csv" | head -1)
    echo "-- $c" >> $O/summary.txt
    for k in mha_fwd_split mha_bwd_split_dkv mha_bwd_split_dq mha_bwd_dq_from_ds mha_bwd_delta mha_bwd_split_cross; do python3 tools/pmc_summary.py $C $k 2>/dev/null | sed "s/^/$k  /" >> $O/summary.txt; done
  done
done
cat $O/summary.txt
find $O -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
