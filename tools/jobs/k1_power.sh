#!/bin/bash
O=gpurun_out/$1; mkdir -p $O
python tools/k1_power_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/power.txt
for v in ${2//,/ }; do echo "== $v"; TSG_HIP_LIB=tools/_ablate/$v.so python tools/k1_power_probe.py 2>&1 | grep -v amdgpu.ids; done | tee -a $O/power.txt
