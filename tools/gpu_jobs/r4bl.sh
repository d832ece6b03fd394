#!/bin/bash
# GPU job of round 4 (bl): K1g backward, what running the row phase of one wave under the column phase of its SIMD partner could give (timing-only ablation)
O=gpurun_out/r4bl; rm -rf $O; mkdir -p $O
for i in 1 2; do TSG_ABL_MASKS=0,512,1536,528,544,1584 python tools/k1_bwd_ablate.py 128 2>&1 | grep "^mask" >> $O/k1g_bwd_overlap_ablation.txt; done
(timeout 600 python -m pytest tests/test_scdm_gpu.py -x -q -m gpu 2>&1 | tail -2) > $O/pytest_scdm.txt
cat $O/k1g_bwd_overlap_ablation.txt $O/pytest_scdm.txt
