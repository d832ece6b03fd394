#!/bin/bash
# memory-side counters of the wgrad kernel (run through gpurun).  Output: gpurun_out/prof_wgrad/mem.txt
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=$PWD/gpurun_out/prof_wgrad; mkdir -p $O
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCC_TAG_STALL_sum TCC_BUSY_sum"; do
  d=$O/m_$(echo $c | tr ' ' '_')
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -o p -- python3 tools/wgrad_only.py 4 > /dev/null 2>&1
  C=$(find $d -name "*counter_collection.csv" | head -1)
  echo "== $c" >> $O/mem.txt
  python3 tools/pmc_summary.py $C wgrad_split >> $O/mem.txt 2>&1
done
cat $O/mem.txt
find $O -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
