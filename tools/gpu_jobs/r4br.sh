#!/bin/bash
# K1 / K1g backward partner exchange: poisoned-workspace stress on the final build (the kernel source was re-arranged for the overlap ablation)
O=gpurun_out/r4br; rm -rf $O; mkdir -p $O
timeout 1500 python tools/k1_bwd_stress.py > $O/k1g_bwd_poisoned_ws_stress.txt 2>&1
grep -v amdgpu $O/k1g_bwd_poisoned_ws_stress.txt | tail -12
