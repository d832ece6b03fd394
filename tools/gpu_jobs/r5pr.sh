#!/bin/bash
O=gpurun_out/r5pr; mkdir -p $O
(timeout 2400 python -m pytest tests/test_lstm_gpu.py tests/test_lstm_soak_gpu.py tests/test_scdm_gpu.py tests/test_models_gpu.py -q -m gpu 2>&1 | grep "passed\|failed\|^E " | head -6) > $O/pytest.txt; cat $O/pytest.txt
python bench.py --no-alt --cpu-sample 0 --no-micro --graph on 2>/dev/null | tail -1 | cut -c1-200
