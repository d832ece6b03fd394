#!/bin/bash
O=gpurun_out/r5h2; mkdir -p $O
(timeout 900 python -m pytest tests/test_models_gpu.py -q -m gpu -x 2>&1 | grep "^E\|passed\|failed\|^tests.*Error\|^FAILED" | head -14) > $O/pytest.txt
cat $O/pytest.txt
