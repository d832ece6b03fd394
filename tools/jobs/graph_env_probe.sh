#!/bin/bash
# The replay defect's reproducer (tests/test_models_gpu.py -k replay_gradients) under runtime switches
O=gpurun_out/$1; mkdir -p $O
for e in "X=1" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "AMD_SERIALIZE_KERNEL=3" "HIP_LAUNCH_BLOCKING=1" "GPU_MAX_HW_QUEUES=1" "DEBUG_HIP_GRAPH_DOT_PRINT=0 HIP_FORCE_DEV_KERNARG=0"; do
  echo "== $e" >> $O/env.txt
  env $e python -m pytest tests/test_models_gpu.py -q -m gpu -k 'replay_gradients' 2>&1 | grep "deviates\|passed\|failed" | cut -c1-200 >> $O/env.txt
done
cat $O/env.txt
