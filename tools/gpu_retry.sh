#!/bin/bash
# developer tool: run a job script on the GPU box through gpurun, retrying while no slot is free.   usage: gpu_retry.sh TIMEOUT script.sh
t=$1; shift
for attempt in 1 2 3 4 5 6 7 8 9 10 11 12; do
  out=$(/usr/local/graft/bin/gpurun --timeout $t -- "bash $*" 2>&1)
  if echo "$out" | grep -q "status=transient"; then sleep 120; continue; fi
  echo "$out"; exit 0
done
echo "$out"; exit 3
