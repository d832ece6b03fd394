#!/bin/bash
O=gpurun_out/r5h2; mkdir -p $O
(timeout 600 python -m pytest tests/test_scdm_gpu.py -q -m gpu -x -k one_node 2>&1 | grep "differ\|passed\|failed" | head -5) > $O/pytest.txt
cat $O/pytest.txt
