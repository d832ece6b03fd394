#!/bin/bash
# GPU job of round 4 (m): the LSTM-layer weight gradient: feature ablation by shape, then memory-side counters of the layer shape
mkdir -p gpurun_out/r4m
python tools/wgrad_lstm_probe.py > gpurun_out/r4m/probe.txt 2>&1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4m
cd $R
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "GRBM_GUI_ACTIVE"; do
  for case in 0 7; do
    d=$O/m_${case}_$(echo $c | tr ' ' '_')
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -o p -- python3 tools/wgrad_lstm_probe.py $case > /dev/null 2>&1
    C=$(find $d -name "*counter_collection.csv" | head -1)
    echo "== case $case: $c" >> $O/pmc.txt
    python3 tools/pmc_summary.py $C wgrad_split >> $O/pmc.txt 2>&1
  done
done
find $O -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
cat $O/probe.txt $O/pmc.txt
