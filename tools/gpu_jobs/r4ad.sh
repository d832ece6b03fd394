#!/bin/bash
mkdir -p gpurun_out/r4ad
cd /tmp && export TMPDIR=/tmp
python3 $GRAFT_REPO_ROOT/tools/profile_r4_k1_traffic.py $GRAFT_REPO_ROOT/gpurun_out/r4ad/k1_pmc_traffic.json 2>&1 | tail -3
