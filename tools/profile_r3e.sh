#!/bin/bash
# PMC traffic of the bf16-storage K1g forward after it moved to the role-specialised kernel (scdm_fwd_ws_kernel<.., bf16_t>, dtype 1).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=$PWD/gpurun_out/prof_r3e; rm -rf $O; mkdir -p $O
for Bp in 128 64; do
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_1_$Bp -o p -- python3 tools/k1_fwd_only.py $Bp 8 1 > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_1_$Bp -o p -- python3 tools/k1_fwd_only.py $Bp 8 1 > /dev/null 2>&1
done
for d in $(cd $O; ls -d pmc_*); do
  C=$(find $O/$d -name "*counter_collection.csv" | head -1)
  echo "== $d" >> $O/k1_ws_bf16_pmc_summary.txt
  python3 tools/pmc_summary.py $C scdm_fwd_ws_kernel 2>/dev/null | sed "s/^/scdm_fwd_ws_kernel  /" >> $O/k1_ws_bf16_pmc_summary.txt
done
cat $O/k1_ws_bf16_pmc_summary.txt
find $O -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
