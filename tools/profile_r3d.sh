#!/bin/bash
# Round-3 PMC session for the role-specialised K1g forward (scdm_fwd_ws_kernel, dtype TSG_F32S): HBM traffic (FETCH_SIZE / WRITE_SIZE
# in separate passes, 128 and 64 pairs per launch) and the SQ / MFMA counters.  Writes gpurun_out/prof_r3d/k1_ws_pmc_summary.txt.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=$PWD/gpurun_out/prof_r3d; rm -rf $O; mkdir -p $O
for Bp in 128 64; do
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_2_$Bp -o p -- python3 tools/k1_fwd_only.py $Bp 8 2 > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_2_$Bp -o p -- python3 tools/k1_fwd_only.py $Bp 8 2 > /dev/null 2>&1
done
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/pmc_sq_2 -o p -- python3 tools/k1_fwd_only.py 128 6 2 > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma_2 -o p -- python3 tools/k1_fwd_only.py 128 6 2 > /dev/null 2>&1
for d in $(cd $O; ls -d pmc_*); do
  C=$(find $O/$d -name "*counter_collection.csv" | head -1)
  echo "== $d" >> $O/k1_ws_pmc_summary.txt
  python3 tools/pmc_summary.py $C scdm_fwd_ws_kernel 2>/dev/null | sed "s/^/scdm_fwd_ws_kernel  /" >> $O/k1_ws_pmc_summary.txt
done
cat $O/k1_ws_pmc_summary.txt
find $O -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
